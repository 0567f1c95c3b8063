#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_exp17; mkdir -p $O
export PYTHONUNBUFFERED=1
ROUNDS=5 STEPS=4 timeout 900 python tools/step_ab.py "base:" "c32:pool_chunk=32" "c128:pool_chunk=128" "r8:pool_rows=8" "r8c32:pool_rows=8,pool_chunk=32" "r8c128:pool_rows=8,pool_chunk=128" "base2:" 2>&1 | grep -v amdgpu.ids | tee $O/step_ab.log
