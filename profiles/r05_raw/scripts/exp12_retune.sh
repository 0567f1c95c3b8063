#!/bin/bash
# Round 5, experiment 12: re-tune what the 24-bit residual stream and the persistent form may have shifted
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_exp12; mkdir -p $O
export PYTHONUNBUFFERED=1
ROUNDS=5 STEPS=4 timeout 900 python tools/step_ab.py "base:" "o0:gemm_variant_o=0" "o_persist:gemm_persist=13" "wo_persist:gemm_persist=25" "group4:gemm_group_m=4" "group16:gemm_group_m=16" "notail:gemm_tail_split=0" "pool128:pool_chunk=128" "base2:" 2>&1 | grep -v amdgpu.ids | tee $O/step_ab.log
