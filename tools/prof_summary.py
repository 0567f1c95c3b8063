"""Top kernels of a rocprofv3 --kernel-trace --stats run: python tools/prof_summary.py <output dir> [n]"""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.3f} ms")
for r in rows[:n]:
    print(f'{r["Name"][:120]:120s} n={r["Calls"]:>6s} avg={float(r["AverageNs"]) / 1e3:9.1f}us tot={float(r["TotalDurationNs"]) / 1e6:9.3f}ms {100 * float(r["TotalDurationNs"]) / tot:5.1f}%')
