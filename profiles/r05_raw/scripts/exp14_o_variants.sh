#!/bin/bash
# Round 5, experiment 14: attention-out projection on two co-resident pipelined 128 x 128 x 64 workgroups per CU (variant 29)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_exp14; mkdir -p $O
export PYTHONUNBUFFERED=1
ONLY=o SKINNY=0 FUSED=1 VARIANTS=26,0,29 ROUNDS=4 timeout 600 python tools/gemm_bench.py 70144 2>&1 | grep -v amdgpu.ids | tee $O/gemm_bench_o.log
ROUNDS=5 STEPS=4 timeout 900 python tools/step_ab.py "base:" "o29:gemm_variant_o=29" "o0:gemm_variant_o=0" "base2:" 2>&1 | grep -v amdgpu.ids | tee $O/step_ab.log
