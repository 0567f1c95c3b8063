#!/bin/bash
# the driver's GPU test command + smoke, logged
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_tests; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee $O/smoke.log
