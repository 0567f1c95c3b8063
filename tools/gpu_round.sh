#!/bin/bash
# One GPU-box session: parity tests (per file, so a crash in one does not hide the others),
# smoke, bench.  Everything is logged under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
KEXPR="${KEXPR:-}"
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
: > gpurun_out/pytest.log
for f in test_kernels_gpu test_fp8_gpu test_encoder_gpu test_edge_cases_gpu test_retriever_gpu test_cli_gpu test_train_forward; do
  echo "=== $f" | tee -a gpurun_out/pytest.log
  if [ -n "$KEXPR" ]; then
    timeout 1200 python -m pytest tests/$f.py -m gpu -q --tb=short -s -k "$KEXPR" 2>&1 | tail -150 | tee -a gpurun_out/pytest.log
  else
    timeout 1200 python -m pytest tests/$f.py -m gpu -q --tb=short -s 2>&1 | tail -150 | tee -a gpurun_out/pytest.log
  fi
done
echo "=== all at once (the driver's command)" | tee -a gpurun_out/pytest.log
timeout 1800 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5 | tee -a gpurun_out/pytest.log
echo "=== smoke" | tee -a gpurun_out/pytest.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -20 | tee gpurun_out/smoke.log
echo "=== bench" 
timeout 900 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 2>&1 | tail -30 | tee gpurun_out/bench.log
