#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_exp8; mkdir -p $O
export PYTHONUNBUFFERED=1
ROUNDS=5 STEPS=4 timeout 900 python tools/step_ab.py "base:" "pc64:pool_chunk=64" "pc32:pool_chunk=32" "ent:embed_nt=1" "pc64_ent:pool_chunk=64,embed_nt=1" "base2:" 2>&1 | grep -v amdgpu.ids | tee $O/step_ab.log
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_kernels_gpu.py tests/test_train_forward.py tests/test_train_step_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.log
