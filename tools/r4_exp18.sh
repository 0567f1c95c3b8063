#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --tb=short -k "attention" 2>&1 | tail -4
for a in 4 8 16; do echo "== att_waves $a"; RP_OPTIONS="att_waves=$a" python tools/attn_bench.py 2>&1 | grep -v amdgpu | cut -c1-80; done
ROUNDS=4 STEPS=3 timeout 600 python tools/step_ab.py "waves4:att_waves=4" "waves8:att_waves=8" "waves16:att_waves=16" 2>&1 | tail -3
