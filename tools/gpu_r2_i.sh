#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
FUSED=1 ONLY=wo VARIANTS=26 STAGGER=0,60,0,60,45,75 ROUNDS=8 timeout 300 python tools/gemm_bench.py 70144 2>&1 | tail -8 | tee gpurun_out/gemm_i.log
for st in 0 60; do RP_OPTIONS=gemm_stagger_us_wo=$st timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --headline-only 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stagger', $st, d['ms_per_step'], d['kernel_ms_per_step']['gemm_wo'])"; done | tee -a gpurun_out/gemm_i.log
echo "=== 2-rank functional run of the N>1 bench path (gloo, one device)" | tee -a gpurun_out/gemm_i.log
RP_BENCH_SHARE_GPU=1 RP_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --headline-only 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['ms_per_step'], d['config']['all_counts_eq_k'], d['config']['sharded_merge_equals_single_gpu'])" | tee -a gpurun_out/gemm_i.log
