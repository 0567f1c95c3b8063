#!/bin/bash
# Round 5, experiment 4: persistent workgroups (one per CU, next tile's first k-tile requested under the epilogue)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_exp4; mkdir -p $O
export PYTHONUNBUFFERED=1
SKINNY=0 FUSED=1 VARIANTS=26 PERSIST=0,29 ROUNDS=4 timeout 600 python tools/gemm_bench.py 70144 2>&1 | grep -v amdgpu.ids | tee $O/gemm_bench_70144.log
ROUNDS=5 STEPS=4 timeout 900 python tools/step_ab.py "base:" "p_wi:gemm_persist=8" "p_wo:gemm_persist=16" "p_qkv:gemm_persist=1" "p_o:gemm_persist=4,gemm_variant_o=26" "o26:gemm_variant_o=26" \
  "p_all:gemm_persist=29" "p_all_notail:gemm_persist=29,gemm_tail_split=0" "base2:" 2>&1 | grep -v amdgpu.ids | tee $O/step_ab.log
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.log
