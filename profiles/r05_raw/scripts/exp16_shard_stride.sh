#!/bin/bash
# Round 5, experiment 16: sampling stride at the 8- / 4-GPU shard shapes (RP_EXPERIMENTS build: scan_stride)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_exp16; mkdir -p $O
export PYTHONUNBUFFERED=1
N=16250 BS=2048 FP8=0,1 IMPLS=0 DENSE=0 CASES="scan_stride=0|scan_stride=2|scan_stride=4|scan_stride=8" timeout 600 python tools/scan_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/scan_8gpu_shape.log
N=32500 BS=1024 FP8=0 IMPLS=0 DENSE=0 CASES="scan_stride=0|scan_stride=2|scan_stride=4|scan_stride=8" timeout 600 python tools/scan_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/scan_4gpu_shape.log
N=130000 BS=256 FP8=0 IMPLS=0 DENSE=0 CASES="scan_stride=0|scan_stride=8|scan_stride=16|scan_stride=32" timeout 600 python tools/scan_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/scan_c2.log
