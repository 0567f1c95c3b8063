#!/bin/bash
# The retrieval-side GPU tests, then rp_sim_topk timings (tools/scan_bench.py: whole call, scan and select classes) at the
# per-rank shapes of an 8- / 4- / 1-GPU step and for a single query, bf16 and e4m3 index.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1 TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_retriever_gpu.py tests/test_fp8_gpu.py tests/test_edge_cases_gpu.py tests/test_shared_index_gpu.py -x -q -m gpu 2>&1 | tail -4
for shape in "2048 16250" "1024 32500" "256 130000" "1 130000"; do
set -- $shape
N=$2 BS=$1 FP8=0,1 IMPLS=0 DENSE=0 timeout 120 python tools/scan_bench.py 2>&1 | grep -v amdgpu.ids | grep -E "B=" | cut -c1-150
done
