"""rp_sim_topk / rp_sim_topk_fp8 micro-benchmark on the GPU box.
env: N (130000), D (1472), BS ("256,128,1"), FP8 ("0,1": which index dtypes), CFGS (scan_cfg values, "0")."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from reprover_amd import _lib
import hip_helpers as hh
if os.environ.get("RP_LIB"):  # a probe / experiment build instead of the product library
    _lib.LIB_PATH = os.path.abspath(os.environ["RP_LIB"])
lib = _lib.load()
N, D, k = int(os.environ.get("N", 130000)), int(os.environ.get("D", 1472)), 100
Bs = [int(b) for b in os.environ.get("BS", "256,128,1").split(",")]
cfgs = [int(c) for c in os.environ.get("CFGS", "0").split(",")]
impls = [int(c) for c in os.environ.get("IMPLS", "0,1").split(",")]  # scan_impl: 0 = pipelined filter kernel, 1 = first generation
# CASES="name=val,name=val|name=val|..." : extra rp_set_option settings, one timing line per case (reset to the option's default afterwards)
cases = [c for c in os.environ.get("CASES", "").split("|")] if os.environ.get("CASES") is not None else [""]
dense_too = os.environ.get("DENSE", "1") == "1"
modes = [int(c) for c in os.environ.get("FP8", "0,1").split(",")]
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(0)
E = torch.empty(N, D, dtype=torch.bfloat16, device=dev)
for lo in range(0, N, 100000):  # chunked: a 1M x 1536 fp32 temporary would be 6 GB
    hi = min(N, lo + 100000)
    E[lo:hi] = torch.nn.functional.normalize(torch.randn(hi - lo, D, generator=g, device=dev), dim=1).to(torch.bfloat16)
E8, es = hh.quantize_e4m3(E)
blocked_modes = [0]
rng = np.random.default_rng(0)
for B in Bs:
    Q = torch.nn.functional.normalize(torch.randn(B, D, generator=g, device=dev), dim=1).to(torch.bfloat16)
    Q8, qs = hh.quantize_e4m3(Q)
    m, acc = hh.synth_masks(rng, N, B, 5000)
    f, ek, bt, own, qk = hh.masks_to_device(m, dev)
    for fp8 in modes:
        for flags in ((0, 1) if dense_too else (0,)):
            for cfg in [(c, i, cs, bl) for c in cfgs for i in (impls if not flags else [0]) for cs in (cases if not flags else [""]) for bl in blocked_modes]:
                cfg, impl, case, bl = cfg
                Ex, E8x, fl = E, E8, flags
                _lib.check(lib.rp_set_option(b"scan_cfg", cfg), "opt")
                _lib.check(lib.rp_set_option(b"scan_impl", impl), "opt")
                for kv in filter(None, case.split(",")):
                    _lib.check(lib.rp_set_option(kv.split("=")[0].encode(), int(kv.split("=")[1])), "opt")
                out_s = torch.empty((B, k), dtype=torch.float32, device=dev); out_i = torch.empty((B, k), dtype=torch.int32, device=dev)
                out_c = torch.empty((B,), dtype=torch.int32, device=dev)
                nb = lib.rp_sim_topk_workspace_bytes(B, N, D, k, flags); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
                def run():
                    if fp8:
                        _lib.check(lib.rp_sim_topk_fp8(Q8.data_ptr(), qs.data_ptr(), E8x.data_ptr(), es.data_ptr(), B, N, D, f.data_ptr(),
                                                       ek.data_ptr(), bt.data_ptr(), bt.shape[0], own.data_ptr(), qk.data_ptr(), 0, k, fl,
                                                       out_s.data_ptr(), out_i.data_ptr(), out_c.data_ptr(), ws.data_ptr(), nb,
                                                       _lib.current_stream()), "sim8")
                    else:
                        _lib.check(lib.rp_sim_topk(Q.data_ptr(), Ex.data_ptr(), B, N, D, f.data_ptr(), ek.data_ptr(), bt.data_ptr(), bt.shape[0],
                                                   own.data_ptr(), qk.data_ptr(), 0, k, fl, out_s.data_ptr(), out_i.data_ptr(), out_c.data_ptr(),
                                                   ws.data_ptr(), nb, _lib.current_stream()), "sim")
                for _ in range(3): run()
                torch.cuda.synchronize()
                it = 20
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(it): run()
                e1.record(); torch.cuda.synchronize()
                tot = e0.elapsed_time(e1) / it   # un-instrumented wall time per call
                _lib.profile_enable(True)        # second loop: per-class split (event pairs around every launch)
                for _ in range(it): run()
                torch.cuda.synchronize()
                prof = _lib.profile_read(); _lib.profile_enable(False)
                byts = N * D * (1 if fp8 else 2) + N * (16 if fp8 else 12)
                scan_s = (prof['scan'][0] + prof['scan_sample'][0]) / it * 1e-3
                print(f"B={B:4d} N={N} D={D} {'e4m3' if fp8 else 'bf16'} {'DENSE' if flags else 'AUTO '} scan_cfg={cfg} impl={impl}: total {tot*1e3:8.1f} us  "
                      f"scan {scan_s*1e6:8.1f} us (sample {prof['scan_sample'][0]/it*1e3:6.1f} + rest {prof['scan'][0]/it*1e3:6.1f})  select {prof['select'][0]/it*1e3:7.1f} us   "
                      f"E-stream {byts/scan_s/1e9:7.1f} GB/s  MFMA {2.0*B*N*D/scan_s/1e12:6.1f} TFLOP/s  QPS {B/(tot*1e-3):10.0f}  "
                      f"cnt_ok {bool((out_c == k).all())}  [{case}]{' BLOCKED' if bl else ''} chk {int(out_i.sum())}", flush=True)
                for kv in filter(None, case.split(",")):
                    _lib.check(lib.rp_set_option(kv.split("=")[0].encode(), {"scan_small_tiles": 1, "scan_waves": 8}.get(kv.split("=")[0], 0)), "opt")
