#!/bin/bash
# Round 5, experiment 18: the round's launch forms on ByT5-base (d_model 1536, 18 layers, d_ff 3968, 12 heads)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_exp18; mkdir -p $O
export PYTHONUNBUFFERED=1
MODEL=byt5-base ROUNDS=4 STEPS=3 timeout 900 python tools/step_ab.py "base:" "persist0:gemm_persist=0" "lds:gemm_rs_lds=1" "r04_forms:gemm_persist=0,gemm_rs_lds=1" "persist_all:gemm_persist=29" "base2:" 2>&1 | grep -v amdgpu.ids | tee $O/step_ab.log
