#!/bin/bash
# Round 5, experiment 11: what bounds the few-token GEMM (64 x 128 x 64 tile, pipelined loop)?  probe builds without the
# fragment reads / without the operand DMA (wrong results, timing only), weights from HBM (COLD=48) and from L2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_exp11; mkdir -p $O
export PYTHONUNBUFFERED=1
for a in "" _NOREADS _NODMA _HALFDMA; do
  for cold in 0 48; do
    echo "=== probe lib '$a' COLD=$cold" | tee -a $O/small_gemm.log
    LIB=tools/probes/_build/libreprover_probe$a.so COLD=$cold FUSED=1 VARIANTS=17 ROUNDS=3 timeout 300 python tools/gemm_bench.py 256 2>&1 | grep -v amdgpu.ids | sed 's/group_m.*: best/best/' | tee -a $O/small_gemm.log
  done
done
