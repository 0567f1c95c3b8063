#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out/r4_exp3; mkdir -p $O
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_kernels_gpu.py -m gpu -q -x --tb=short 2>&1 | tail -25 > $O/pytest.log
tail -5 $O/pytest.log
ROUNDS=5 STEPS=3 timeout 600 python tools/step_ab.py "rowscale_launches:gemm_rs_lds=0" "rs_lds:" > $O/step_ab.log 2>&1
tail -3 $O/step_ab.log
