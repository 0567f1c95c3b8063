#!/bin/bash
# PMC passes over any command: bash tools/kernel_pmc.sh <kernel-name-substring> <out-tag> -- <command...>
# prints, per (kernel, grid size), the average counter values per dispatch
KFILT="$1"; TAG="$2"; shift 3
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out/pmc_$TAG; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
           "SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  (cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $grp -d $O/p$i -o a --output-format csv -- "$@" > $O/p$i.log 2>&1)
done
KFILT="$KFILT" O="$O" python - <<'PY'
import csv, glob, collections, os
O, K = os.environ["O"], os.environ["KFILT"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if K not in r["Kernel_Name"]: continue
        agg[(r["Kernel_Name"][:70], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, d in sorted(agg.items()):
    print(key, {k: round(sum(v) / len(v)) for k, v in sorted(d.items())}, "n=", max(len(v) for v in d.values()))
PY
