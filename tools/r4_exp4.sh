#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out/r4_exp4; mkdir -p $O
FUSED=1 SKINNY=0 VARIANTS=16,17,18,19 ROUNDS=4 timeout 600 python tools/gemm_bench.py 256 > $O/gemm_bench_128.log 2>&1
grep -v amdgpu.ids $O/gemm_bench_128.log
