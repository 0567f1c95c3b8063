#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out/r4_exp12; mkdir -p $O
ROUNDS=5 STEPS=3 timeout 600 python tools/step_ab.py "default:" "rs_rowscale:gemm_rs_lds=0" 2>&1 | tail -3
timeout 600 python bench.py --steps 10 --warmup 3 --headline-only --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], {k: round(v,3) for k,v in d['kernel_ms_per_step'].items() if v})"
