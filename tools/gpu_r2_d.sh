#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
echo "=== tests" | tee gpurun_out/pytest_d.log
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py tests/test_edge_cases_gpu.py -m gpu -q --tb=short -x -k "topk or shard or fp8" 2>&1 | tail -12 | tee -a gpurun_out/pytest_d.log
echo "=== scan cases" | tee gpurun_out/scan_d.log
DENSE=0 IMPLS=0,1 BS=256 FP8=0 timeout 300 python tools/scan_bench.py 2>&1 | tail -30 | tee -a gpurun_out/scan_d.log
cd /tmp && export TMPDIR=/tmp
DENSE=0 IMPLS=0 BS=256 FP8=0 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_scan -o scan --output-format csv -- python $GRAFT_REPO_ROOT/tools/scan_bench.py > $GRAFT_REPO_ROOT/gpurun_out/prof_scan.log 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_scan/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:5]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:8.1f}')
PY
