#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out/r4_exp11; mkdir -p $O
echo skip-tests
timeout 900 python bench.py --steps 5 --warmup 2 --no-train-step 2>&1 | tail -12 > $O/bench.log
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4_exp11/bench.log") if l.startswith('{"metric"')][-1])
for k in ("value", "ms_per_step", "roofline_hbm", "roofline_b1", "shard_call_us", "b1_latency_ms", "scan_only_qps", "scan_only_qps_e4m3_index", "product_api_qps", "leg_errors"):
    print(k, json.dumps(d.get(k))[:900])
print("roofline_scan", json.dumps(d["roofline_scan"])[:900])
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:1200])
PY
