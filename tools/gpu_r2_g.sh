#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
( time timeout 1200 python bench.py ) 2>&1 | tail -8 | cut -c1-6000 | tee gpurun_out/bench_g.log
