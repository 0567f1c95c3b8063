#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
echo "=== tests" | tee gpurun_out/pytest_c.log
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py tests/test_retriever_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -30 | tee -a gpurun_out/pytest_c.log
echo "=== scan cases" | tee gpurun_out/scan_c.log
DENSE=1 IMPLS=0 BS=256,1 FP8=0 BLOCKED=0,1 CASES="|scan_filter_cfg=1|scan_no_epilogue=1|scan_filter_cfg=1,scan_no_epilogue=1|scan_sample_cfg=1" timeout 300 python tools/scan_bench.py 2>&1 | tail -30 | tee -a gpurun_out/scan_c.log
