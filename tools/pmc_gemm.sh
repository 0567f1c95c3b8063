#!/bin/bash
# PMC counters for the GEMM variants (separate passes; --pmc only with --kernel-trace).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "^\s*(SQ_|TCC_|TCP_|GRBM_|TA_)[A-Z_0-9]+" -o | sort -u | head -400 > gpurun_out/pmc/counters.txt
wc -l gpurun_out/pmc/counters.txt
run() { # name, counters
  rocprofv3 --kernel-trace --pmc $2 -d gpurun_out/pmc/$1 -o out --output-format csv -- env VARIANTS=${VARIANTS:-8,10} ONLY=wi python tools/gemm_bench.py > gpurun_out/pmc/$1.log 2>&1
}
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES"
run sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT"
find gpurun_out/pmc -name "*counter_collection.csv" | head
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc/*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        if "gemm" not in k: continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])] += 1
    print("==", f)
    for k, d in agg.items():
        print(k)
        for c, v in d.items(): print(f"   {c:36s} {v/cnt[(k,c)]:16.1f} (avg of {cnt[(k,c)]})")
PY
