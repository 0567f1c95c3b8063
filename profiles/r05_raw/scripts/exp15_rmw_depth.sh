#!/bin/bash
# Round 5, experiment 15: blocks of old x values in flight in the residual epilogue of the 8-wave tile (2 = libprev, 3, 4)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_exp15; mkdir -p $O
export PYTHONUNBUFFERED=1
for i in 1 2; do
LIB=profiles/r05_raw/scripts/libprev.so ROUNDS=4 STEPS=4 timeout 900 python tools/step_ab.py "depth2:" 2>&1 | grep median | tee -a $O/step_ab.log
ROUNDS=4 STEPS=4 timeout 900 python tools/step_ab.py "depth3:" 2>&1 | grep median | tee -a $O/step_ab.log
LIB=profiles/r05_raw/scripts/libdepth4.so ROUNDS=4 STEPS=4 timeout 900 python tools/step_ab.py "depth4:" 2>&1 | grep median | tee -a $O/step_ab.log
done
