#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1 TMPDIR=/tmp; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_retriever_gpu.py tests/test_fp8_gpu.py tests/test_edge_cases_gpu.py tests/test_shared_index_gpu.py -x -q -m gpu 2>&1 | tail -3
for shape in "2048 16250" "1024 32500" "512 65000"; do
set -- $shape
N=$2 BS=$1 FP8=0,1 IMPLS=0 DENSE=0 timeout 120 python tools/scan_bench.py 2>&1 | grep -v amdgpu.ids | grep -E "B=" | cut -c1-150
done
