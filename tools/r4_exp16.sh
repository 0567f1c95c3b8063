#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_retriever_gpu.py tests/test_kernels_gpu.py tests/test_fp8_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x --tb=short 2>&1 | tail -25
