cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6s2
for v in "" _halfdma _nodma _sametile; do
  echo "=== probe lib '$v'" 
  RP_LIB=tools/probes/_build/libreprover_probe$v.so BS=256 FP8=0 IMPLS=0 DENSE=0 CASES="|scan_no_epilogue=1" timeout 300 python tools/scan_bench.py 2>&1 | grep -v Warning
done
echo "=== product lib"
BS=256 FP8=0 IMPLS=0 DENSE=0 CASES="|scan_no_epilogue=1" timeout 300 python tools/scan_bench.py 2>&1 | grep -v Warning
