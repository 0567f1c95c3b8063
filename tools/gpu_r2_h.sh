#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python tools/latency_bench.py 2>&1 | grep -v amdgpu | tail -4 | tee gpurun_out/lat_h.log
RP_OPTIONS=small_t_schedule=0 timeout 600 python tools/latency_bench.py 2>&1 | grep -v amdgpu | tail -4 | tee -a gpurun_out/lat_h.log
cd /tmp && export TMPDIR=/tmp
NBYTES=100 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_lat -o lat --output-format csv -- python $GRAFT_REPO_ROOT/tools/latency_bench.py > $GRAFT_REPO_ROOT/gpurun_out/prof_lat.log 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_lat/**/*kernel_stats.csv", recursive=True)
tot = 0
for r in list(csv.DictReader(open(f[0])))[:14]:
    if "rp::" in r["Name"] and int(r["Calls"]) >= 50:
        per = float(r["TotalDurationNs"]) / 1e3 / 55   # 55 retrieve calls (5 warm-up + 25 + 25)
        tot += per
        print(f'{r["Name"][:80]:80s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:7.1f} us/call {per:7.1f}')
print("sum us per retrieve:", round(tot, 1))
PY
