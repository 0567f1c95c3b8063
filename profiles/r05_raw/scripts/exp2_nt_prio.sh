#!/bin/bash
# Round 5, experiment 2: cache policy of the epilogues' global accesses, static wave priority
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_exp2; mkdir -p $O
export PYTHONUNBUFFERED=1
ROUNDS=5 STEPS=4 timeout 900 python tools/step_ab.py "base:" "nt1:gemm_epi_nt=1" "nt2:gemm_epi_nt=2" "nt4:gemm_epi_nt=4" "nt6:gemm_epi_nt=6" "nt7:gemm_epi_nt=7" \
  "prio1:gemm_prio=1" "prio2:gemm_prio=2" "nt7prio1:gemm_epi_nt=7,gemm_prio=1" "base2:" 2>&1 | tee $O/step_ab.log
