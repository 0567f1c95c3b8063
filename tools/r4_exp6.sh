#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out/r4_exp6; mkdir -p $O
echo "--- weights from HBM (rotating copies)" > $O/gemm_cold.log
COLD=48 FUSED=1 SKINNY=0 VARIANTS=17,15,0 ROUNDS=3 RP_OPTIONS="gemm_small_pipe=0" ONLY=wo timeout 300 python tools/gemm_bench.py 256 2>&1 | grep -v amdgpu.ids | sed 's/variant=17/variant=17(plain16)/' >> $O/gemm_cold.log
for o in wo qk "o " wi; do
COLD=48 FUSED=1 SKINNY=0 VARIANTS=16,18,19 ROUNDS=3 ONLY="$o" timeout 300 python tools/gemm_bench.py 256 2>&1 | grep -v amdgpu.ids >> $O/gemm_cold.log
done
cut -c1-150 $O/gemm_cold.log
