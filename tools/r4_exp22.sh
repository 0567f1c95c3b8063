#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1
for v in 1 2 3 1 2 3; do
echo "gemm_small_pipe=$v"; RP_OPTIONS="gemm_small_pipe=$v" timeout 300 python tools/latency_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-75
done
