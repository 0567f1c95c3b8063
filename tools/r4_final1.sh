#!/bin/bash
# round 4: tests, smoke, the full bench line, rocprof evidence
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r4_pytest_all.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/r4_smoke.log
timeout 1200 python bench.py 2>&1 | tail -2 > gpurun_out/r4_bench_full.log
tail -1 gpurun_out/r4_bench_full.log | cut -c1-600
