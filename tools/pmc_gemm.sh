#!/bin/bash
# PMC counters for GEMM variants on the FFN-in shape (separate passes; --pmc only with --kernel-trace).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/pmc/avail.txt 2>&1
grep -oE "\b(SQ|TCC|TCP|GRBM|TA|TD|SPI)_[A-Za-z_0-9]+" gpurun_out/pmc/avail.txt | sort -u > gpurun_out/pmc/counters.txt
wc -l gpurun_out/pmc/counters.txt
run() { # name, counters  (each pass under its own timeout: a rejected counter set can hang rocprofv3)
  rm -rf gpurun_out/pmc/$1
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $2 -d gpurun_out/pmc/$1 -o out --output-format csv -- env VARIANTS=${VARIANTS:-26,20} ONLY=${ONLY:-wi} ROUNDS=1 python tools/gemm_bench.py > gpurun_out/pmc/$1.log 2>&1 || echo "pass $1 failed: $(grep -m1 -i 'error code' gpurun_out/pmc/$1.log)"
}
run ta1 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
run ta2 "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum"
run tcp1 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"
run tcp2 "TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
run tcc1 "TCC_HIT_sum TCC_MISS_sum"
run tcc2 "TCC_REQ_sum TCC_TAG_STALL_sum"
run tcc3 "TCC_BUSY_sum TCC_CYCLE_sum"
run lds "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc/*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gemm_kernel" not in k: continue
        k = k[k.index("GemmCfg"):][:40]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])] += 1
    for k, d in agg.items():
        for c, v in d.items(): print(f"{k:42s} {c:40s} {v/cnt[(k,c)]:18.1f} (avg of {cnt[(k,c)]})")
PY
