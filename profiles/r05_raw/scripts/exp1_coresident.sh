#!/bin/bash
# Round 5, experiment 1: two co-resident 4-wave workgroups per CU (GEMM variants 27 / 28) against the 8-wave 256 x 256 tile.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_exp1; mkdir -p $O
export PYTHONUNBUFFERED=1
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu.txt
SKINNY=0 FUSED=1 VARIANTS=26,27,28 ROUNDS=4 timeout 600 python tools/gemm_bench.py 70144 2>&1 | tee $O/gemm_bench_70144.log
ROUNDS=5 STEPS=4 timeout 900 python tools/step_ab.py "base:" "wi27:gemm_variant=27" "wi28:gemm_variant=28" "wo27:gemm_variant_wo=27" "wo28:gemm_variant_wo=28" \
  "qkv27:gemm_variant_qkv=27" "qkv28:gemm_variant_qkv=28" "o27:gemm_variant_o=27" "o28:gemm_variant_o=28" \
  "all27:gemm_variant=27,gemm_variant_wo=27,gemm_variant_qkv=27,gemm_variant_o=27" \
  "all28:gemm_variant=28,gemm_variant_wo=28,gemm_variant_qkv=28,gemm_variant_o=28" "base2:" 2>&1 | tee $O/step_ab.log
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.log
