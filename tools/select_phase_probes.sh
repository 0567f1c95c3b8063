#!/bin/bash
# Phase split of select_kernel / gather_select_kernel at the 8-GPU shard shape: a probe build (RP_EXPERIMENTS=1 python -m
# reprover_amd.build, after touching a source) returns from the kernels early by bits of the scan_no_epilogue option:
#   8 gather: at once | 16 gather: no run entries | 32 gather: no predicate | 64 gather: before the select body
#   256 select: list loaded | 512 after the first barrier | 1024 after the radix passes | 2048 after the sort
# Bits that skip the SAMPLE stage's outputs (256 ... 2048 on the sample select) leave the filter pass without a bound: only
# at a small shape (few rows per query) does the call stay short - do NOT run them at 256 x 130,000 (minutes per call).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1 TMPDIR=/tmp; mkdir -p gpurun_out
for dbg in 0 264 576 1024 2048 16 32 64; do
echo "== B=2048 N=16250 scan_no_epilogue=$dbg"
rm -rf gpurun_out/prof_shard
N=16250 BS=2048 FP8=0 IMPLS=0 DENSE=0 CASES="scan_no_epilogue=$dbg" timeout 60 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_shard -o s --output-format csv -- python tools/scan_bench.py > /dev/null 2>&1
python tools/prof_summary.py gpurun_out/prof_shard 2>&1 | grep -E "select" | cut -c1-50,115-150
done
