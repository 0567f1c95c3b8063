#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py -m gpu -q -x --tb=short -k "attention or golden or agree" 2>&1 | tail -4
python tools/attn_bench.py 2>&1 | grep -v amdgpu | cut -c1-80
ROUNDS=4 STEPS=3 timeout 600 python tools/step_ab.py "now:" 2>&1 | tail -1
