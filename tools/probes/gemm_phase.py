"""Where does a GEMM workgroup spend its time?  Builds a probe copy of the library with
-DRP_PHASE_PROBE (here, on CPU) or loads it (on the GPU box) and prints per-phase times.

  python tools/probes/gemm_phase.py build        # here: cross-compile the probe library
  VARIANTS=6,20 python tools/probes/gemm_phase.py   # on the GPU box
"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
# the probe library lives beside this tool, NOT in the product's lib/ (it ships to the GPU box only while it exists here:
# delete tools/probes/_build/ when done)
PROBE = os.path.join(ROOT, "tools", "probes", "_build", "libreprover_probe%s.so" % (("_" + os.environ["ABLATE"]) if os.environ.get("ABLATE") else ""))
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(os.path.dirname(PROBE), exist_ok=True)
    src = [os.path.join(ROOT, "reprover_amd", "csrc", f) for f in ("rp_encoder.hip", "rp_retrieval.hip", "rp_train.hip", "rp_comm.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-DRP_PHASE_PROBE", "-DRP_EXPERIMENTS", *os.environ.get("ABLATE_DEFS", "").split(), *src, "-o", PROBE])
    print("built", PROBE); sys.exit(0)
import numpy as np, torch
from reprover_amd import _lib
_lib.LIB_PATH = PROBE
lib = _lib.load()
lib.rp_probe_read_phase_ts.argtypes = [C.c_void_p, C.c_int]
lib.rp_probe_read_handover_ts.argtypes = [C.c_void_p]
dev = torch.device("cuda")
M = int(os.environ.get("M", 65536))
shapes = {"wi": (7168, 1472, _lib.RP_EPI_GEGLU_BF16), "wo": (1472, 3584, _lib.RP_EPI_RESID),
          "o": (1472, 384, _lib.RP_EPI_RESID),
          "qkv": (1152, 1472, _lib.RP_EPI_STORE_BF16)}
for name in os.environ.get("ONLY", "wi,wo").split(","):
    N, K, epi = shapes[name]
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    out = (torch.zeros(2, M, N, dtype=torch.bfloat16, device=dev) if epi == _lib.RP_EPI_RESID else
           torch.empty(M, N // 2 if epi == _lib.RP_EPI_GEGLU_BF16 else N, dtype=torch.bfloat16, device=dev))
    persist = int(os.environ.get("PERSIST", "0"))
    for v in [int(x) for x in os.environ.get("VARIANTS", "6,20").split(",")]:
        _lib.check(lib.rp_set_option(b"gemm_variant_all", v), "opt")
        _lib.check(lib.rp_set_option(b"gemm_persist", persist), "opt")
        if persist:  # persistent workgroups keep SUMS per workgroup: read them around a batch of launches
            def launch():
                if epi == _lib.RP_EPI_RESID:
                    ssp_ = torch.empty((N + 63) // 64, M, device=dev)
                    _lib.check(lib.rp_dbg_gemm_fused(A.data_ptr(), W.data_ptr(), out.data_ptr(), M, N, K, N, epi, None, 0, 0.0,
                                                     0.0, None, ssp_.data_ptr(), (N + 63) // 64, _lib.current_stream()), "gemm")
                else:
                    _lib.check(lib.rp_dbg_gemm(A.data_ptr(), W.data_ptr(), out.data_ptr(), M, N, K, N, epi, _lib.current_stream()), "gemm")
            for _ in range(3):
                launch()
            torch.cuda.synchronize()
            b = np.zeros(4 * 256, dtype=np.uint64); a_ = np.zeros(4 * 256, dtype=np.uint64)
            assert lib.rp_probe_read_phase_ts(b.ctypes.data, b.size) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                launch()
            e1.record(); torch.cuda.synchronize()
            assert lib.rp_probe_read_phase_ts(a_.ctypes.data, a_.size) == 0
            d = (a_.astype(np.int64) - b.astype(np.int64)).reshape(256, 4)
            n = d[:, 3].astype(np.float64)
            span = e0.elapsed_time(e1) / 10 * 1e3
            pro, main, ep = d[:, 0] / n / 100.0, d[:, 1] / n / 100.0, d[:, 2] / n / 100.0
            print(f"{name} variant {v} PERSISTENT: launch {span:.1f} us, tiles per workgroup {n.mean() / 10:.2f}")
            print(f"   per tile: top-of-loop -> first k-tile ready {pro.mean():6.2f} us | main loop {main.mean():6.2f} us "
                  f"({main.mean() / (K / 64):.3f} per 64-K step) | epilogue + end barrier {ep.mean():6.2f} us | total {(pro + main + ep).mean():6.2f}")
            print(f"   sum of tile times / (256 CUs x launch) = {((d[:, 0] + d[:, 1] + d[:, 2]).sum() / 100.0 / 10) / (256 * span):.3f}")
            continue
        fused = epi == _lib.RP_EPI_RESID
        np_ = (N + 63) // 64
        ssp = torch.empty(np_, M, device=dev) if fused else None
        for _ in range(12):
            if fused:
                _lib.check(lib.rp_dbg_gemm_fused(A.data_ptr(), W.data_ptr(), out.data_ptr(), M, N, K, N, epi, None, 0, 0.0,
                                                 0.0, None, ssp.data_ptr(), np_, _lib.current_stream()), "gemm")
            else:
                _lib.check(lib.rp_dbg_gemm(A.data_ptr(), W.data_ptr(), out.data_ptr(), M, N, K, N, epi, _lib.current_stream()), "gemm")
        torch.cuda.synchronize()
        tiles = ((N + 255) // 256) * (M // 256)
        ts = np.zeros(4 * tiles, dtype=np.uint64)
        assert lib.rp_probe_read_phase_ts(ts.ctypes.data, ts.size) == 0
        ts = ts.reshape(tiles, 4).astype(np.int64)
        t0 = ts[:, 0].min()
        us = lambda x: x / 100.0  # 100 MHz
        pro, main, epi_t = us(ts[:, 1] - ts[:, 0]), us(ts[:, 2] - ts[:, 1]), us(ts[:, 3] - ts[:, 2])
        span = us(ts[:, 3].max() - t0)
        order = np.argsort(ts[:, 0])
        first = order[:256]; later = order[256:]
        print(f"{name} variant {v}: tiles {tiles}, kernel span {span:.1f} us, K-tiles(64) {K // 64}")
        for lab, idx in (("first wave of WGs", first), ("later WGs", later)):
            print(f"   {lab:18s}: prologue {pro[idx].mean():6.2f} us | main loop {main[idx].mean():6.2f} us "
                  f"({main[idx].mean() / (K / 64):.3f} us per 64-K step; p10 {np.percentile(main[idx],10):.2f} p90 {np.percentile(main[idx],90):.2f}) | "
                  f"epilogue {epi_t[idx].mean():6.2f} us | total {(pro+main+epi_t)[idx].mean():6.2f}")
        if v == 20:
            h = np.zeros(8 * 64 * 3, dtype=np.uint64)
            assert lib.rp_probe_read_handover_ts(h.ctypes.data) == 0
            h = h.reshape(8, 64, 3).astype(np.int64)
            nk = min(K // 64 - 1, 64)
            for w in range(4):
                hw = h[w, :nk]
                step = np.diff(hw[:, 0])
                print(f"   WG 1000 wave {w}: cycles per K-tile median {np.median(step):.0f}; vmcnt wait median "
                      f"{np.median(hw[:,1]-hw[:,0]):.0f} max {np.max(hw[:,1]-hw[:,0])}; barrier wait median "
                      f"{np.median(hw[:,2]-hw[:,1]):.0f} max {np.max(hw[:,2]-hw[:,1])}")
        busy = (pro + main + epi_t).sum() / 256 / span
        print(f"   sum of WG lifetimes / (256 CUs x span) = {busy:.3f}")
