#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out/r4_exp8; mkdir -p $O
for h in 24 48 64 96 128; do
RP_OPTIONS="gemm_helpers=$h" NBYTES=100,300,600 REPEAT=2 timeout 600 python tools/latency_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-48 | sort | awk -v h=$h '{print "helpers", h, $0}'
done | tee $O/latency_helpers.log
