#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "variant" 2>&1 | tail -4
for o in wo qk wi "o "; do
FUSED=1 SKINNY=0 VARIANTS=17,18,19 ROUNDS=3 ONLY="$o" RP_OPTIONS="gemm_small_pipe=0" timeout 300 python tools/gemm_bench.py 256 2>&1 | grep -v amdgpu.ids | cut -c1-150
done
