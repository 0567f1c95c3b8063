#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
echo "=== tests" | tee gpurun_out/pytest_d.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -x -k "sim_topk or shard" 2>&1 | tail -8 | tee -a gpurun_out/pytest_d.log
echo "=== scan cases" | tee gpurun_out/scan_d.log
DENSE=0 IMPLS=0 BS=256 FP8=0 CASES="|scan_no_epilogue=4|scan_no_epilogue=1|scan_filter_cfg=3|scan_filter_cfg=3,scan_sample_cfg=1" timeout 300 python tools/scan_bench.py 2>&1 | tail -30 | tee -a gpurun_out/scan_d.log
