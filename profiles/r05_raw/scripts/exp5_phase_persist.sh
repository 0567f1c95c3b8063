#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_exp5; mkdir -p $O
export PYTHONUNBUFFERED=1
for P in 0 29; do
PERSIST=$P M=70144 ONLY=wi,wo,o,qkv VARIANTS=26 timeout 300 python tools/probes/gemm_phase.py 2>&1 | grep -v amdgpu.ids | tee -a $O/phase.log
done
