"""Where does a workgroup of the similarity scan's filter pass spend its time?  Probe build only
(python tools/probes/gemm_phase.py build; then on the GPU box:  FILTER_CFGS=0,3 python tools/probes/scan_phase.py).
Thread 0 of every filter workgroup stamps the 100 MHz wall clock at its start / first k-tile landed / main loop done /
epilogue done (RP_TS, probes/rp_probe_hooks.h); the workgroups are grouped by the round of the chip they ran in."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from reprover_amd import _lib
_lib.LIB_PATH = os.environ.get("RP_LIB", os.path.join(ROOT, "tools", "probes", "_build", "libreprover_probe.so"))
import hip_helpers as hh
lib = _lib.load()
lib.rp_probe_read_scan_phase_ts.argtypes = [C.c_void_p, C.c_int]
N, D, k, B = int(os.environ.get("N", 130000)), int(os.environ.get("D", 1472)), 100, int(os.environ.get("B", 256))
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(0)
E = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1).to(torch.bfloat16)
Q = torch.nn.functional.normalize(torch.randn(B, D, generator=g, device=dev), dim=1).to(torch.bfloat16)
rng = np.random.default_rng(0)
m, acc = hh.synth_masks(rng, N, B, 5000)
f, ek, bt, own, qk = hh.masks_to_device(m, dev)
out_s = torch.empty((B, k), dtype=torch.float32, device=dev); out_i = torch.empty((B, k), dtype=torch.int32, device=dev)
out_c = torch.empty((B,), dtype=torch.int32, device=dev)
nb = lib.rp_sim_topk_workspace_bytes(B, N, D, k, 0); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
def run():
    _lib.check(lib.rp_sim_topk(Q.data_ptr(), E.data_ptr(), B, N, D, f.data_ptr(), ek.data_ptr(), bt.data_ptr(), bt.shape[0], own.data_ptr(),
                               qk.data_ptr(), 0, k, 0, out_s.data_ptr(), out_i.data_ptr(), out_c.data_ptr(), ws.data_ptr(), nb,
                               _lib.current_stream()), "sim")
blocks = (N + 255) // 256
ref = None
for cfg in [int(c) for c in os.environ.get("FILTER_CFGS", "0").split(",")]:
    _lib.check(lib.rp_set_option(b"scan_filter_cfg", cfg), "opt")
    for _ in range(5): run()
    torch.cuda.synchronize()
    ids = out_i.cpu().numpy().copy(); sc = out_s.cpu().numpy().copy()
    if ref is None: ref = (ids, sc)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ts = np.zeros(4 * 1024, dtype=np.uint64)
    assert lib.rp_probe_read_scan_phase_ts(ts.ctypes.data, ts.size) == 0
    ts = ts.reshape(-1, 4).astype(np.int64)
    live = ts[:, 3] > ts[:, 0]
    # the filter pass wrote the stamps of its workgroups last: keep those whose four stamps are ordered and recent
    t_end = ts[live, 3].max()
    sel = live & (ts[:, 0] > t_end - 100 * 400)  # within 400 us of the end
    t = ts[sel]; t0 = t[:, 0].min()
    us = lambda x: x / 100.0
    pro, main, epi = us(t[:, 1] - t[:, 0]), us(t[:, 2] - t[:, 1]), us(t[:, 3] - t[:, 2])
    start = us(t[:, 0] - t0)
    order = np.argsort(start)
    first, later = order[:256], order[256:]
    print(f"scan_filter_cfg={cfg}: whole call {e0.elapsed_time(e1) / 20 * 1e3:.1f} us; filter workgroups {sel.sum()}, span {us(t[:, 3].max() - t0):.1f} us; "
          f"same ids as cfg 0: {bool((ids == ref[0]).all())}, same score bits: {bool((sc.view(np.int32) == ref[1].view(np.int32)).all())}")
    for lab, idx in (("first 256", first), ("later", later)):
        if len(idx) == 0: continue
        print(f"   {lab:10s}: start {start[idx].mean():6.1f} (p10 {np.percentile(start[idx], 10):.1f} p90 {np.percentile(start[idx], 90):.1f}) | "
              f"first k-tile {pro[idx].mean():5.2f} | main loop {main[idx].mean():6.2f} (p10 {np.percentile(main[idx], 10):.2f} p90 {np.percentile(main[idx], 90):.2f}) | "
              f"epilogue {epi[idx].mean():5.2f} (p90 {np.percentile(epi[idx], 90):.2f}) | end {us(t[idx, 3] - t0).mean():6.1f} (max {us(t[idx, 3] - t0).max():.1f})")
_lib.check(lib.rp_set_option(b"scan_filter_cfg", 0), "opt")
