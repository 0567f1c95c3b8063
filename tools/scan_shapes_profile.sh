#!/bin/bash
# rp_sim_topk kernel by kernel at the per-rank shapes of an 8- / 4- / 1-GPU step (N x 256 queries against 130,000 / N rows):
# tools/scan_bench.py under rocprofv3 --kernel-trace --stats, top kernels per shape (profiles/r04_select_stages.md).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1 TMPDIR=/tmp; mkdir -p gpurun_out
for shape in ${SHAPES:-"2048 16250" "1024 32500" "256 130000"}; do
set -- $shape
echo "== B=$1 N=$2"
rm -rf gpurun_out/prof_shard
N=$2 BS=$1 FP8=${FP8:-0} IMPLS=0 DENSE=0 timeout 120 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_shard -o s --output-format csv -- python tools/scan_bench.py 2>&1 | grep -v amdgpu.ids | grep -E "B=" | cut -c1-160
python tools/prof_summary.py gpurun_out/prof_shard 2>&1 | head -6
done
