#!/bin/bash
# rocprofv3 kernel-trace + stats of the training-step leg (tools/train_bench.py): per-kernel times -> gpurun_out/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=${ROUND:-r03}; B=${BATCH:-8}
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train_$R -o train --output-format csv -- \
  python tools/train_bench.py $B > gpurun_out/prof_train_$R.log 2>&1
tail -1 gpurun_out/prof_train_$R.log | cut -c1-300
R=$R B=$B python - <<'PY'
import csv, glob, os
R, B = os.environ["R"], os.environ["B"]
f = glob.glob(f"gpurun_out/prof_train_{R}/*kernel_stats.csv")
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = ["kernel,calls,total_ms,avg_us,pct"]
for r in rows[:40]:
    out.append(f'"{r["Name"][:120]}",{r["Calls"]},{float(r["TotalDurationNs"])/1e6:.3f},{float(r["AverageNs"])/1e3:.1f},{100*float(r["TotalDurationNs"])/tot:.2f}')
open(f"gpurun_out/{R}_train_step_b{B}_kernel_stats.csv", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
