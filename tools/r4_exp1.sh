#!/bin/bash
# round 4, GPU session 1: new GEMM tile forms (28/29/30/31, touch), deep filter ring - correctness, then timings
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out/r4_exp1; mkdir -p $O
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --tb=short 2>&1 | tail -25 > $O/pytest_kernels.log
tail -3 $O/pytest_kernels.log
FUSED=1 SKINNY=0 VARIANTS=26,20,30,31,28,29 TOUCH=0,1 ROUNDS=3 timeout 600 python tools/gemm_bench.py 70144 > $O/gemm_bench_70144.log 2>&1
tail -40 $O/gemm_bench_70144.log
ROUNDS=4 STEPS=3 timeout 600 python tools/step_ab.py \
  "r03:gemm_exact_n=0,gemm_touch=0" \
  "touch_only:gemm_exact_n=0,gemm_touch=1" \
  "exact_only:gemm_exact_n=1,gemm_touch=0" \
  "new_default:" \
  "wo30_wi30:gemm_exact_n=0,gemm_variant=30,gemm_variant_wo=30,gemm_variant_qkv=30" \
  "wi30_only:gemm_variant=30" \
  "wi31_only:gemm_variant=31" \
  "wo20:gemm_exact_n=0,gemm_variant_wo=20" \
  "o_small:gemm_variant_o=0" \
  "group16:gemm_group_m=16" \
  > $O/step_ab.log 2>&1
tail -14 $O/step_ab.log
DENSE=0 BS=256 IMPLS=0 CASES="scan_deep=1|scan_deep=0|scan_deep=1|scan_deep=0" timeout 300 python tools/scan_bench.py > $O/scan_c2.log 2>&1
cat $O/scan_c2.log | cut -c1-330
N=16250 DENSE=1 BS=2048 FP8=0 IMPLS=0 CASES="scan_deep=1|scan_deep=0" timeout 300 python tools/scan_bench.py > $O/scan_shard8.log 2>&1
cat $O/scan_shard8.log | cut -c1-330
N=1000000 D=1536 DENSE=0 BS=256 IMPLS=0 CASES="scan_deep=1|scan_deep=0" timeout 300 python tools/scan_bench.py > $O/scan_c5.log 2>&1
cat $O/scan_c5.log | cut -c1-330
