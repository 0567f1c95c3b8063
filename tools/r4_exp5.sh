#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out/r4_exp5; mkdir -p $O
for M in 512 1024 2048; do
FUSED=1 SKINNY=0 VARIANTS=16,17,0,26 ROUNDS=3 timeout 600 python tools/gemm_bench.py $M 2>&1 | grep -v amdgpu.ids >> $O/gemm_small.log
done
cut -c1-150 $O/gemm_small.log
NBYTES=100,300,1000 REPEAT=2 timeout 600 python tools/latency_bench.py 2>&1 | grep -v amdgpu.ids > $O/latency_pipe.log
RP_OPTIONS="gemm_small_pipe=0" NBYTES=100,300,1000 REPEAT=2 timeout 600 python tools/latency_bench.py 2>&1 | grep -v amdgpu.ids > $O/latency_plain.log
echo PIPE; cut -c1-400 $O/latency_pipe.log; echo PLAIN; cut -c1-400 $O/latency_plain.log
