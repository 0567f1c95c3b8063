#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out/r4_exp9; mkdir -p $O
: > $O/gemm.log
for o in wo qk wi; do
FUSED=1 SKINNY=0 VARIANTS=17,18,19 ROUNDS=3 ONLY="$o" timeout 300 python tools/gemm_bench.py 256 2>&1 | grep -v amdgpu.ids >> $O/gemm.log
COLD=48 FUSED=1 SKINNY=0 VARIANTS=17,18,19 ROUNDS=3 ONLY="$o" timeout 300 python tools/gemm_bench.py 256 2>&1 | grep -v amdgpu.ids | sed 's/^/COLD /' >> $O/gemm.log
done
cut -c1-160 $O/gemm.log
