#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
ROUND=r04 bash tools/profile_round.sh 2>&1 | tail -60 | cut -c1-260
CASES=r04 NBYTES=100 bash tools/latency_profile.sh 2>&1 | cut -c1-200 | tee gpurun_out/r04_latency_profile.txt
