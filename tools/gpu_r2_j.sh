#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_edge_cases_gpu.py tests/test_retriever_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -5
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --headline-only 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'])"
