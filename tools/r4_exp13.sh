#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out/r4_exp13; mkdir -p $O
timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_kernels_gpu.py -m gpu -q -x --tb=short 2>&1 | tail -8
DENSE=0 BS=256 IMPLS=0 timeout 300 python tools/scan_bench.py 2>&1 | grep "^B=" | cut -c1-330 | tee $O/scan_c2.log
