cd $GRAFT_REPO_ROOT
FILTER_CFGS=4,0,4,0 timeout 300 python tools/probes/scan_phase.py 2>&1 | grep -v "Warning\|amdgpu.ids"
echo "=== scan_bench product build"
BS=256 FP8=0,1 IMPLS=0 DENSE=0 timeout 300 python tools/scan_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids"
N=16250 BS=2048 FP8=0,1 IMPLS=0 DENSE=0 timeout 300 python tools/scan_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids"
echo "=== tests"
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_retriever_gpu.py tests/test_fp8_gpu.py -x -q -m gpu 2>&1 | tail -15
