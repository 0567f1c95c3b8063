#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_exp10; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py tests/test_retriever_gpu.py -m gpu -q -x 2>&1 | tail -12 | tee $O/pytest.log
for i in 1 2; do
LIB=profiles/r05_raw/scripts/libreprover_prev.so ROUNDS=4 STEPS=4 timeout 900 python tools/step_ab.py "two_bf16_planes:" 2>&1 | grep median | tee -a $O/step_ab.log
ROUNDS=4 STEPS=4 timeout 900 python tools/step_ab.py "bf16+int8_planes_rne:" 2>&1 | grep median | tee -a $O/step_ab.log
done
