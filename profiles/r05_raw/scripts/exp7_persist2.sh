#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_exp7; mkdir -p $O
export PYTHONUNBUFFERED=1
ROUNDS=5 STEPS=4 timeout 900 python tools/step_ab.py "base:" "p_wi:gemm_persist=8" "p_wi_nolds:gemm_persist=8,gemm_rs_lds=0" "nolds:gemm_rs_lds=0" "p_wi_qkv_nolds:gemm_persist=9,gemm_rs_lds=0" "p_wi_wo:gemm_persist=24" "base2:" 2>&1 | grep -v amdgpu.ids | tee $O/step_ab.log
