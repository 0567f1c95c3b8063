#!/bin/bash
# PMC passes over tools/attn_bench.py (attention kernel alone): where its wave-cycles go
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/attn_pmc; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --pmc $grp -d $O/p$i -o a --output-format csv -- python $GRAFT_REPO_ROOT/tools/attn_bench.py > $O/p$i.log 2>&1)
done
python - <<'PY'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/attn_pmc"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attention_kernel" not in r["Kernel_Name"]: continue
        agg[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for grid, d in sorted(agg.items(), key=lambda kv: int(kv[0])):
    print("grid", grid, {k: round(sum(v) / len(v)) for k, v in sorted(d.items())})
PY
