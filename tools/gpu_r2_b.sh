#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
echo "=== tests" | tee gpurun_out/pytest_b.log
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py tests/test_retriever_gpu.py -m gpu -q --tb=short -s 2>&1 | tail -40 | tee -a gpurun_out/pytest_b.log
echo "=== scan cases" | tee gpurun_out/scan_b.log
DENSE=0 IMPLS=0 BS=256 FP8=0 CASES="|scan_filter_cfg=1|scan_no_epilogue=1|scan_filter_cfg=1,scan_no_epilogue=1|scan_sample_cfg=1|scan_stride=32|scan_stride=32,scan_filter_cfg=1|scan_stride=8,scan_filter_cfg=1" timeout 300 python tools/scan_bench.py 2>&1 | tail -12 | tee -a gpurun_out/scan_b.log
DENSE=0 IMPLS=0 BS=1 FP8=0 CASES="|scan_filter_cfg=1|scan_no_epilogue=1|scan_filter_cfg=1,scan_no_epilogue=1" timeout 300 python tools/scan_bench.py 2>&1 | tail -6 | tee -a gpurun_out/scan_b.log
