#!/bin/bash
# Round 5, experiment 3: what bounds the epilogues?  per-workgroup phase timestamps with parts of the epilogue removed
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_exp3; mkdir -p $O
export PYTHONUNBUFFERED=1
for a in "" NOSTORE NOGELU NOLOAD NOLOADSTORE; do
  echo "=== ablation: ${a:-none}" | tee -a $O/phase.log
  ABLATE=$a M=70144 ONLY=wi,wo,o,qkv VARIANTS=26 timeout 300 python tools/probes/gemm_phase.py 2>&1 | grep -v amdgpu.ids | tee -a $O/phase.log
done
