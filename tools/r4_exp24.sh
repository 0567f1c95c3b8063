#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_train_step_gpu.py -x -q -m gpu 2>&1 | tail -5
for pl in 1 0 1 0; do
echo "pipelined=$pl"; RP_TRAIN_PIPELINED=$pl timeout 300 python tools/train_bench.py 8 64 2>&1 | grep -v amdgpu.ids | grep -o '"batch[0-9]*": {"ms_per_step": [0-9.]*'
done
