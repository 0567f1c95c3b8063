"""BASELINE.json configs[4] at full size on one MI355X (a supplementary measurement; the contract
bench is bench.py on configs[1]): ByT5-base encoder (random init) over 256 states of the benchmark's
length mix + masked top-100 over 1,000,000 synthetic premises held as an e4m3 index.
    python tools/c5_bench.py [--steps 5]   ->   one JSON line (also the bf16-index scan for comparison)
"""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util
from reprover_amd import _lib, synth, tokenizer
from reprover_amd.common import Fp8Index
from reprover_amd.encoder import HipT5Encoder
import hip_helpers as hh

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); _argv = sys.argv; sys.argv = ["x"]; spec.loader.exec_module(bench); sys.argv = _argv
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=5); args = ap.parse_args()
dev = torch.device("cuda"); lib = _lib.load()
cfg = synth.t5_config("byt5-base"); D = cfg["d_model"]
enc = HipT5Encoder(cfg, bench.random_init_state_dict(cfg, dev, 1), dev, torch.bfloat16)
N, B, k, F = 1_000_000, 256, 100, 5000
g = torch.Generator(device=dev); g.manual_seed(5)
codes = torch.empty((N, D), dtype=torch.uint8, device=dev); scale = torch.empty((N,), device=dev)
E16 = torch.empty((N, D), dtype=torch.bfloat16, device=dev)
for lo in range(0, N, 125_000):
    x = torch.nn.functional.normalize(torch.randn(125_000, D, generator=g, device=dev), dim=1)
    E16[lo:lo + 125_000] = x.to(torch.bfloat16)
    c, s = hh.quantize_e4m3(x); codes[lo:lo + 125_000], scale[lo:lo + 125_000] = c, s
rng = np.random.default_rng(5)
lens = synth.synth_lengths(rng, B, "mix", lo=16, hi=2048)
ids_np, cu_np = tokenizer.encode_packed([synth.synth_state(rng, int(n) - 1) for n in lens], 2048)
T, max_len = int(cu_np[-1]), int(np.diff(cu_np).max())
ids_d, cu_d = torch.from_numpy(ids_np).to(dev), torch.from_numpy(cu_np).to(dev)
m, acc = hh.synth_masks(rng, N, B, F)
f, ek, bt, own, qk = hh.masks_to_device(m, dev)
q = torch.empty((B, D), dtype=torch.bfloat16, device=dev)
out_s = torch.empty((B, k), device=dev); out_i = torch.empty((B, k), dtype=torch.int32, device=dev)
out_c = torch.empty((B,), dtype=torch.int32, device=dev)
nb = lib.rp_sim_topk_workspace_bytes(B, N, D, k, 0); ws = torch.empty(nb, dtype=torch.uint8, device=dev)

def step(fp8=True):
    enc.encode_packed_device(ids_d, cu_d, B, T, max_len, q)
    if fp8:
        q8 = Fp8Index.quantize(q)
        _lib.check(lib.rp_sim_topk_fp8(q8.codes.data_ptr(), q8.scale.data_ptr(), codes.data_ptr(), scale.data_ptr(), B, N, D,
                                       f.data_ptr(), ek.data_ptr(), bt.data_ptr(), bt.shape[0], own.data_ptr(), qk.data_ptr(), 0, k, 0,
                                       out_s.data_ptr(), out_i.data_ptr(), out_c.data_ptr(), ws.data_ptr(), nb, _lib.current_stream()), "fp8")
    else:
        _lib.check(lib.rp_sim_topk(q.data_ptr(), E16.data_ptr(), B, N, D, f.data_ptr(), ek.data_ptr(), bt.data_ptr(), bt.shape[0],
                                   own.data_ptr(), qk.data_ptr(), 0, k, 0, out_s.data_ptr(), out_i.data_ptr(), out_c.data_ptr(),
                                   ws.data_ptr(), nb, _lib.current_stream()), "bf16")

res = {}
for name, fp8 in (("e4m3_index", True), ("bf16_index", False)):
    for _ in range(2): step(fp8)
    torch.cuda.synchronize(); _lib.profile_enable(True); t0 = time.perf_counter()
    for _ in range(args.steps): step(fp8)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
    prof = _lib.profile_read(); _lib.profile_enable(False)
    Tp = (T + 255) // 256 * 256
    wi_ms, wi_n = prof["gemm_wi"]
    scan_ms = (prof["scan"][0] + prof["scan_sample"][0]) / args.steps
    res[name] = {"ms_per_step": dt * 1e3, "queries_per_s": B / dt, "counts_eq_k": bool((out_c == k).all()),
                 "scan_ms": scan_ms, "select_ms": prof["select"][0] / args.steps,
                 "scan_hbm_GBps": N * D * (1 if fp8 else 2) / (scan_ms * 1e-3) / 1e9,
                 "scan_mfma_TFLOPs": 2.0 * B * N * D / (scan_ms * 1e-3) / 1e12,
                 "ffn_in_gemm_TFLOPs": 2.0 * Tp * D * 2 * cfg["d_ff"] * wi_n / (wi_ms * 1e-3) / 1e12}
print(json.dumps({"workload": "configs[4]: ByT5-base encode of 256 states (%d byte-tokens) + masked top-100 over 1,000,000 x 1536 premises" % T,
                  "n_gpus": 1, "data": "synthetic", **res}))
