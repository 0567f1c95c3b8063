cd $GRAFT_REPO_ROOT
echo "=== scan_bench product build: persistent vs one workgroup per tile"
BS=256 FP8=0 IMPLS=0 DENSE=0 CASES="|scan_persist=0||scan_persist=0||scan_persist=0" timeout 300 python tools/scan_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids"
N=16250 BS=2048 FP8=0 IMPLS=0 DENSE=0 CASES="|scan_persist=0||scan_persist=0" timeout 300 python tools/scan_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids"
N=32500 BS=1024 FP8=0 IMPLS=0 DENSE=0 CASES="|scan_persist=0||scan_persist=0" timeout 300 python tools/scan_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids"
echo "=== sample pass with / without its epilogue (probe build)"
RP_LIB=tools/probes/_build/libreprover_probe.so BS=256 FP8=0 IMPLS=0 DENSE=0 CASES="|scan_no_epilogue=64||scan_no_epilogue=64" timeout 300 python tools/scan_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids"
echo "=== tests"
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_retriever_gpu.py tests/test_fp8_gpu.py -x -q -m gpu 2>&1 | tail -5
