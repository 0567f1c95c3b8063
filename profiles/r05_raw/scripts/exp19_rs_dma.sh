#!/bin/bash
# Round 5, experiment 19: the tile's 256 RMSNorm factors by one LDS-DMA piece (RowScaleDma) against a global read at epilogue entry
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_exp19; mkdir -p $O
export PYTHONUNBUFFERED=1
ROUNDS=5 STEPS=4 timeout 900 python tools/step_ab.py "dma:" "global:gemm_rs_dma=0" "dma_nopersist:gemm_persist=0" "global_nopersist:gemm_rs_dma=0,gemm_persist=0" "dma_persist_wi_only:gemm_persist=8" "dma2:" 2>&1 | grep -v amdgpu.ids | tee $O/step_ab.log
timeout 600 python -m pytest tests/test_encoder_gpu.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.log
