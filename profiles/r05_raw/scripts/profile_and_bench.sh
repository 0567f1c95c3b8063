#!/bin/bash
# Round-5 evidence: the default bench line (un-profiled), then the rocprofv3 kernel-trace + PMC passes of the headline step,
# then the single-state latency profile.  Judged summaries -> gpurun_out/profiles_r05/ (copied into profiles/ by hand)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/profiles_r05
timeout 1500 python bench.py --steps 20 --warmup 5 2> gpurun_out/bench_r05.err | tail -1 > gpurun_out/profiles_r05/r05_bench_unprofiled.json
tail -2 gpurun_out/bench_r05.err
ROUND=r05 bash tools/profile_round.sh 2>&1 | tail -30
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b1_r05 -o b1 --output-format csv -- env NBYTES=100 REPEAT=1 python tools/latency_bench.py > gpurun_out/prof_b1_r05.log 2>&1
python tools/prof_summary.py gpurun_out/prof_b1_r05 22 > gpurun_out/profiles_r05/r05_b1_latency_kernel_stats.txt 2>&1
tail -3 gpurun_out/prof_b1_r05.log
