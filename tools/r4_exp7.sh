#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out/r4_exp7; mkdir -p $O
: > $O/gemm_cold.log
for h in 0 64 192; do
echo "--- helpers $h" >> $O/gemm_cold.log
RP_OPTIONS="gemm_helpers=$h" COLD=48 FUSED=1 SKINNY=0 VARIANTS=16 ROUNDS=3 timeout 300 python tools/gemm_bench.py 256 2>&1 | grep -v amdgpu.ids >> $O/gemm_cold.log
done
cut -c1-150 $O/gemm_cold.log
for h in 0 192; do
RP_OPTIONS="gemm_helpers=$h" NBYTES=100,300,1000 REPEAT=2 timeout 600 python tools/latency_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-60 > $O/latency_h$h.log
echo "helpers $h"; cat $O/latency_h$h.log
done
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py tests/test_retriever_gpu.py -m gpu -q -x --tb=short 2>&1 | tail -5
