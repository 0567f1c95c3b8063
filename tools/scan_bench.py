"""rp_sim_topk micro-benchmark on the GPU box (BASELINE config 2 shape by default)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from reprover_amd import _lib
import hip_helpers as hh
lib = _lib.load()
N, D, k = int(os.environ.get("N", 130000)), 1472, 100
Bs = [int(b) for b in os.environ.get("BS", "256,128,1").split(",")]
cfgs = [int(c) for c in os.environ.get("CFGS", "0,1,2").split(",")]
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(0)
E = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1).to(torch.bfloat16)
rng = np.random.default_rng(0)
for B in Bs:
    Q = torch.nn.functional.normalize(torch.randn(B, D, generator=g, device=dev), dim=1).to(torch.bfloat16)
    m, acc = hh.synth_masks(rng, N, B, 5000)
    dm = hh.masks_to_device(m, dev)
    for flags in (0, 1):
        for cfg in cfgs:
            _lib.check(lib.rp_set_option(b"scan_cfg", cfg), "opt")
            f, ek, bt, own, qk = dm
            out_s = torch.empty((B, k), dtype=torch.float32, device=dev); out_i = torch.empty((B, k), dtype=torch.int32, device=dev)
            out_c = torch.empty((B,), dtype=torch.int32, device=dev)
            nb = lib.rp_sim_topk_workspace_bytes(B, N, D, k, flags); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
            def run():
                _lib.check(lib.rp_sim_topk(Q.data_ptr(), E.data_ptr(), B, N, D, f.data_ptr(), ek.data_ptr(), bt.data_ptr(), bt.shape[0],
                                           own.data_ptr(), qk.data_ptr(), 0, k, flags, out_s.data_ptr(), out_i.data_ptr(), out_c.data_ptr(),
                                           ws.data_ptr(), nb, _lib.current_stream()), "sim")
            for _ in range(3): run()
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            it = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(it): run()
            e1.record(); torch.cuda.synchronize()
            prof = _lib.profile_read(); _lib.profile_enable(False)
            tot = e0.elapsed_time(e1) / it
            byts = N * D * 2 + N * 12
            print(f"B={B:4d} N={N} {'DENSE' if flags else 'AUTO '} scan_cfg={cfg}: total {tot*1e3:8.1f} us  scan {prof['scan'][0]/it*1e3:8.1f} us "
                  f"({prof['scan'][1]//it} launches)  select {prof['select'][0]/it*1e3:7.1f} us   E-stream {byts/(prof['scan'][0]/it*1e-3)/1e9:7.1f} GB/s "
                  f" QPS {B/(tot*1e-3):10.0f}  cnt_ok {bool((out_c == k).all())}", flush=True)
