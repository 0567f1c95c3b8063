#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out/r4_exp14; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --tb=short -k "sim_topk or shard" 2>&1 | tail -4
DENSE=0 BS=256,64,1 IMPLS=0 FP8=0 timeout 300 python tools/scan_bench.py 2>&1 | grep "^B=" | cut -c1-330 | tee $O/scan.log
