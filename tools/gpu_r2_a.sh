#!/bin/bash
# Round 2, GPU session A: parity of the second-generation scan + first measurements.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
for f in test_kernels_gpu test_fp8_gpu test_edge_cases_gpu test_encoder_gpu test_retriever_gpu test_cli_gpu; do
  echo "=== $f" | tee -a gpurun_out/pytest_a.log
  timeout 900 python -m pytest tests/$f.py -m gpu -q --tb=short -s -x 2>&1 | tail -60 | tee -a gpurun_out/pytest_a.log
done
echo "=== scan bench" | tee gpurun_out/scan_a.log
BS=256,64,1 FP8=0 timeout 300 python tools/scan_bench.py 2>&1 | tail -20 | tee -a gpurun_out/scan_a.log
N=1000000 D=1536 BS=256 FP8=0,1 timeout 300 python tools/scan_bench.py 2>&1 | tail -20 | tee -a gpurun_out/scan_a.log
echo "=== gemm stagger" | tee gpurun_out/gemm_a.log
M=70144
FUSED=1 ONLY=wo VARIANTS=26 STAGGER=0,30,60,90 ROUNDS=4 timeout 300 python tools/gemm_bench.py $M 2>&1 | tail -8 | tee -a gpurun_out/gemm_a.log
FUSED=1 ONLY=wi VARIANTS=20 STAGGER=0,10,20,35 ROUNDS=4 timeout 300 python tools/gemm_bench.py $M 2>&1 | tail -8 | tee -a gpurun_out/gemm_a.log
FUSED=1 ONLY=qkv VARIANTS=26 STAGGER=0,15,30 ROUNDS=4 timeout 300 python tools/gemm_bench.py $M 2>&1 | tail -8 | tee -a gpurun_out/gemm_a.log
FUSED=1 ONLY=wi VARIANTS=20 GROUP_M=2,4,8,16 ROUNDS=3 timeout 300 python tools/gemm_bench.py $M 2>&1 | tail -8 | tee -a gpurun_out/gemm_a.log
echo "=== bench" | tee gpurun_out/bench_a.log
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -3 | tee -a gpurun_out/bench_a.log
