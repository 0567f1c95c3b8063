#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out/r4_exp10; mkdir -p $O
timeout 1500 python -m pytest tests/ -m gpu -q -x --tb=short 2>&1 | tail -8 | tee $O/pytest_all.log
NBYTES=100,300,1000 REPEAT=2 timeout 600 python tools/latency_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-60 | tee $O/latency.log
