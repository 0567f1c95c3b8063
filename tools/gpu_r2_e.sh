#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for f in test_kernels_gpu test_fp8_gpu test_edge_cases_gpu test_encoder_gpu test_retriever_gpu test_cli_gpu; do
  echo "=== $f" | tee -a gpurun_out/pytest_e.log
  timeout 900 python -m pytest tests/$f.py -m gpu -q --tb=short -x 2>&1 | tail -15 | tee -a gpurun_out/pytest_e.log
done
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke_e.log
N=1000000 D=1536 BS=256 FP8=0,1 DENSE=0 IMPLS=0,1 timeout 300 python tools/scan_bench.py 2>&1 | tail -6 | tee gpurun_out/scan_e.log
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1500 | tee gpurun_out/bench_e.log
