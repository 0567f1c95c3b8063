#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1 TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_retriever_gpu.py tests/test_fp8_gpu.py tests/test_edge_cases_gpu.py tests/test_shared_index_gpu.py -x -v -m gpu 2>&1 | grep -v "PASSED" | head -60
