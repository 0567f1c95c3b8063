#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out/r4_exp15; mkdir -p $O
for a in 0 96 192 384; do
RP_OPTIONS="attn_prefetch=$a" NBYTES=100,300,600 REPEAT=2 timeout 600 python tools/latency_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-48 | sort | awk -v h=$a '{print "attn_prefetch", h, $0}'
done | tee $O/latency.log
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_retriever_gpu.py -m gpu -q -x --tb=short 2>&1 | tail -3
