#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1 TMPDIR=/tmp; mkdir -p gpurun_out
for shape in "2048 16250" "256 130000"; do
set -- $shape
for dbg in 0 264 576 1024 2048 16 32; do
echo "== B=$1 N=$2 debug=$dbg"
rm -rf gpurun_out/prof_shard
N=$2 BS=$1 FP8=0 IMPLS=0 DENSE=0 CASES="scan_no_epilogue=$dbg" timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_shard -o s --output-format csv -- python tools/scan_bench.py > /dev/null 2>&1
python tools/prof_summary.py gpurun_out/prof_shard 2>&1 | grep -E "select" | cut -c1-50,115-150
done
done
