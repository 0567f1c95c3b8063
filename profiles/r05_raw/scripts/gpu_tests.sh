#!/bin/bash
# the driver's GPU test command + smoke (+ a short bench), logged
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_tests; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -25 | tee $O/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -6 | tee $O/smoke.log
if [ -n "$BENCH" ]; then timeout 1500 python bench.py 2> $O/bench.err | tail -1 > $O/bench.json; tail -3 $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print({k:d[k] for k in ('value','ms_per_step','scan_only_qps','scan_only_qps_e4m3_index','product_api_qps')}); print(d['roofline']); print(d['cpu_baseline']); print(d['roofline_hbm']); print(d['b1_latency_ms']); print(d.get('leg_errors'))"; fi
