#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1 TMPDIR=/tmp; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_retriever_gpu.py tests/test_fp8_gpu.py tests/test_edge_cases_gpu.py tests/test_shared_index_gpu.py -x -q -m gpu 2>&1 | tail -3
for shape in "2048 16250" "256 130000"; do
set -- $shape
echo "== B=$1 N=$2"
rm -rf gpurun_out/prof_shard
N=$2 BS=$1 FP8=0 IMPLS=0 DENSE=0 timeout 120 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_shard -o s --output-format csv -- python tools/scan_bench.py 2>&1 | grep -v amdgpu.ids | grep -E "B=" | cut -c1-130
python tools/prof_summary.py gpurun_out/prof_shard 2>&1 | grep -E "select|filter|scan_k" | cut -c1-50,115-150
done
