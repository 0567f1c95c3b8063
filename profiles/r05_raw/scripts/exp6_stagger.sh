#!/bin/bash
# Round 5, experiment 6: first-round stagger on the HBM-bound attention-out projection (RP_EXPERIMENTS build)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_exp6; mkdir -p $O
export PYTHONUNBUFFERED=1
ONLY=o SKINNY=0 FUSED=1 VARIANTS=26 STAGGER=0,5,10,15,20,30 ROUNDS=4 timeout 600 python tools/gemm_bench.py 70144 2>&1 | grep -v amdgpu.ids | tee $O/gemm_bench_o.log
ONLY=wo SKINNY=0 FUSED=1 VARIANTS=26 STAGGER=0,15,30,60 ROUNDS=3 timeout 600 python tools/gemm_bench.py 70144 2>&1 | grep -v amdgpu.ids | tee $O/gemm_bench_wo.log
ROUNDS=4 STEPS=4 timeout 900 python tools/step_ab.py "base:gemm_stagger_us_o=0" "o26:gemm_variant_o=26,gemm_stagger_us_o=0" "o26s10:gemm_variant_o=26,gemm_stagger_us_o=10" "o26s20:gemm_variant_o=26,gemm_stagger_us_o=20" "o26s30:gemm_variant_o=26,gemm_stagger_us_o=30" "base2:gemm_stagger_us_o=0" 2>&1 | grep -v amdgpu.ids | tee $O/step_ab.log
