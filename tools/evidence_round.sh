#!/bin/bash
# One GPU-box session for a round's committed evidence, in the order the files depend on each other:
#   1. rocprofv3 stats + the four PMC passes of the bench command (tools/profile_round.sh) -> pmc_traffic.json stamped with
#      the kernel-source hash, copied into profiles/ ON THE BOX so that
#   2. the full bench line of the same box and session quotes `roofline.traffic`;
#   3. configs[4] (bench.py --config c5), the single-state latency kernels, 4. every GPU test.
# Everything lands under gpurun_out/evidence_$ROUND/; copy what is to be judged into profiles/ afterwards.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${ROUND:-r06}; OUT=gpurun_out/evidence_$R; mkdir -p $OUT
ROUND=$R bash tools/profile_round.sh > $OUT/profile_round.log 2>&1
cp gpurun_out/profiles_$R/* $OUT/ 2>/dev/null
cp gpurun_out/profiles_$R/pmc_traffic.json profiles/pmc_traffic.json
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_full.log 2>&1
grep '^{"metric"' $OUT/bench_full.log | tail -1 > $OUT/${R}_bench_unprofiled.json
timeout 900 python bench.py --config c5 --steps 5 > $OUT/c5.log 2>&1
grep '^{"metric"' $OUT/c5.log | tail -1 > $OUT/${R}_c5_byt5base_1m_e4m3.json
CASES=$R NBYTES=100 bash tools/latency_profile.sh > $OUT/${R}_b1_latency_kernel_stats.txt 2>&1
timeout 2400 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -15 $OUT/pytest_gpu.log
cp gpurun_out/parity_margins.json $OUT/${R}_parity_margins.json 2>/dev/null
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
head -c 1500 $OUT/${R}_bench_unprofiled.json; echo; head -c 600 $OUT/${R}_c5_byt5base_1m_e4m3.json; echo
