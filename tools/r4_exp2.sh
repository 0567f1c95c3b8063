#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out/r4_exp2; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --tb=short -k "gemm" 2>&1 | tail -25 > $O/pytest_kernels.log
tail -5 $O/pytest_kernels.log
ROUNDS=5 STEPS=3 timeout 600 python tools/step_ab.py \
  "r03:gemm_exact_n=0,gemm_tail_inlaunch=0" \
  "new_default:" \
  "tail_only:gemm_exact_n=0" \
  "exact_only:gemm_tail_inlaunch=0" \
  "wo26_tail:gemm_exact_n=0,gemm_tail_split=0" \
  "no_tails_at_all:gemm_exact_n=0,gemm_tail_inlaunch=0,gemm_tail_split=0" \
  > $O/step_ab.log 2>&1
tail -8 $O/step_ab.log
