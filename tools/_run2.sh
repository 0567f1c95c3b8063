cd $GRAFT_REPO_ROOT
FILTER_CFGS=0,3,0,3 timeout 300 python tools/probes/scan_phase.py 2>&1 | grep -v "Warning\|amdgpu.ids"
echo "=== scan_bench on the probe build: cfg 0 vs 3"
RP_LIB=tools/probes/_build/libreprover_probe.so BS=256 FP8=0 IMPLS=0 DENSE=0 CASES="|scan_filter_cfg=3||scan_filter_cfg=3|scan_no_epilogue=1|scan_filter_cfg=3,scan_no_epilogue=1" timeout 300 python tools/scan_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids"
echo "=== 8-GPU shard shape"
RP_LIB=tools/probes/_build/libreprover_probe.so N=16250 BS=2048 FP8=0 IMPLS=0 DENSE=0 CASES="|scan_filter_cfg=3||scan_filter_cfg=3" timeout 300 python tools/scan_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids"
