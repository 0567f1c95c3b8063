#!/bin/bash
# Round 5, experiment 13: the embedding as a row copy from the pre-encoded table
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_exp13; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_encoder_gpu.py tests/test_kernels_gpu.py tests/test_train_step_gpu.py tests/test_train_forward.py tests/test_retriever_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee $O/pytest.log
ROUNDS=5 STEPS=4 timeout 900 python tools/step_ab.py "base:" "lds:gemm_rs_lds=1" "base2:" 2>&1 | grep -v amdgpu.ids | tee $O/step_ab.log
