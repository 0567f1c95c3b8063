#!/bin/bash
# rocprofv3 evidence for the bench command: kernel-trace + stats (one run), then PMC passes
# (separate runs, --kernel-trace only, as the guide prescribes).  Summaries -> gpurun_out/prof_*/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=${ROUND:-r05}
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$R -o bench --output-format csv -- \
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --headline-only > gpurun_out/prof_$R.log 2>&1
tail -1 gpurun_out/prof_$R.log | cut -c1-400
python - <<PY
import csv, collections, glob
f = glob.glob("gpurun_out/prof_$R/*kernel_stats.csv")
print(f)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = ["kernel,calls,total_ms,avg_us,pct"]
for r in rows[:25]:
    out.append(f'"{r["Name"][:110]}",{r["Calls"]},{float(r["TotalDurationNs"])/1e6:.3f},{float(r["AverageNs"])/1e3:.1f},{100*float(r["TotalDurationNs"])/tot:.2f}')
open("gpurun_out/prof_${R}_kernel_stats_top.csv","w").write("\n".join(out)+"\n")
print("\n".join(out[:16]))
PY
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -k 5 180 rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/pmc_${R}_$tag -o bench --output-format csv -- \
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/pmc_${R}_$tag.log 2>&1
done
python - <<PY
import csv, collections, glob
res = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/pmc_${R}_*/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "rp::" not in k: continue
        key = (k[:110], row["Counter_Name"])
        agg[key][0] += float(row["Counter_Value"]); agg[key][1] += 1
    for (k, c), (v, n) in agg.items():
        res[k][c] = (v / n, n)
lines = ["kernel,counter,avg_per_launch,launches"]
for k, d in sorted(res.items()):
    for c, (v, n) in sorted(d.items()):
        lines.append(f'"{k}",{c},{v:.1f},{n}')
open("gpurun_out/pmc_${R}_summary.csv", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
PY
# ---- copy the judged summaries into profiles/ (tracked) -----------------------------------------
mkdir -p gpurun_out/profiles_$R
cp gpurun_out/prof_${R}_kernel_stats_top.csv gpurun_out/profiles_$R/${R}_bench_kernel_stats.csv
cp gpurun_out/pmc_${R}_summary.csv gpurun_out/profiles_$R/${R}_bench_pmc_summary.csv
grep "^{\"metric\"" gpurun_out/prof_$R.log | tail -1 > gpurun_out/profiles_$R/${R}_bench_under_rocprof.json
R=$R python - <<'PY'
import csv, json, os
R = os.environ["R"]
rows = list(csv.DictReader(open(f"gpurun_out/pmc_{R}_summary.csv")))
def get(kfrag, counter):
    for r in rows:
        if kfrag in r["kernel"] and r["counter"] == counter:
            return float(r["avg_per_launch"])
    return None
import sys
sys.path.insert(0, ".")
import bench
out = {"kernel_source_hash": bench.kernel_source_hash(),
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB per launch) of 'python bench.py "
                 "--headline-only'; read bytes = 2 x FETCH_SIZE x 1024 (gfx950 wide-read correction, "
                 "MI355X_MICROARCH.md HBM section), write bytes = WRITE_SIZE x 1024; Infinity-Cache hits are included",
       "round": R}
f, w = get("EpiGegluBf16", "FETCH_SIZE"), get("EpiGegluBf16", "WRITE_SIZE")
if f is not None and w is not None:
    out["gemm_wi_bytes_per_launch"] = 2 * f * 1024 + w * 1024
    out["gemm_wi_fetch_kib"], out["gemm_wi_write_kib"] = f, w
fs, ws = get("sim_scan_kernel", "FETCH_SIZE"), get("sim_scan_kernel", "WRITE_SIZE")
ff, wf = get("sim_filter_kernel", "FETCH_SIZE"), get("sim_filter_kernel", "WRITE_SIZE")
if None not in (fs, ws, ff, wf):  # one sample-pass launch + one filter-pass launch per step
    out["scan_bytes_per_step"] = (2 * fs * 1024 + ws * 1024) + (2 * ff * 1024 + wf * 1024)
    out["scan_sample_fetch_kib"], out["scan_sample_write_kib"] = fs, ws
    out["scan_filter_fetch_kib"], out["scan_filter_write_kib"] = ff, wf
json.dump(out, open(f"gpurun_out/profiles_{R}/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out)[:700])
PY
