#!/bin/bash
# rocprofv3 kernel durations of single-state retrieve() calls (graph replay).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for g in ${CASES:-default}; do
  timeout -k 5 300 rocprofv3 --kernel-trace --stats -d gpurun_out/lat_prof_$g -o lat --output-format csv -- \
    env NBYTES=${NBYTES:-100} python tools/latency_bench.py > gpurun_out/lat_prof_$g.log 2>&1
  python - <<PY
import csv, glob
f = glob.glob("gpurun_out/lat_prof_$g/*kernel_stats.csv")
rows = [r for r in csv.DictReader(open(f[0])) if "rp::" in r["Name"]]
print("== $g")
for r in rows[:16]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>6s}  avg {float(r["AverageNs"])/1e3:7.2f} us  total {float(r["TotalDurationNs"])/1e6:8.3f} ms')
PY
done
