"""attention_kernel alone (rp_dbg_attention): uniform-length batches and the bench's length mix.
python tools/attn_bench.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from reprover_amd import _lib, synth
import hip_helpers as hh
lib = _lib.load()
dev = torch.device("cuda:0")
H = 6
g = torch.Generator(device=dev); g.manual_seed(0)
tab = torch.randn(H, 257, generator=g, device=dev)
rng = np.random.default_rng(synth.SEED + 100)
cases = {"mix(256 states)": synth.synth_lengths(rng, 256, "mix", lo=16, hi=2048)}
for L, n in ((64, 1024), (128, 512), (256, 256), (512, 128), (2048, 32)):
    cases[f"{n} x {L}"] = np.full(n, L)
for name, lens in cases.items():
    lens = np.asarray(lens, dtype=np.int64)
    T = int(lens.sum()); Tp = (T + 255) // 256 * 256
    cu = torch.zeros(len(lens) + 1, dtype=torch.int32, device=dev); cu[1:] = torch.from_numpy(np.cumsum(lens)).to(dev)
    qkv = (torch.randn(Tp, 3 * H * 64, generator=g, device=dev) * 0.5).to(torch.bfloat16)
    out = torch.zeros((Tp, H * 64), dtype=torch.bfloat16, device=dev)
    def run():
        _lib.check(lib.rp_dbg_attention(qkv.data_ptr(), cu.data_ptr(), tab.data_ptr(), out.data_ptr(), len(lens), int(lens.max()),
                                        H, Tp, _lib.current_stream()), "att")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    flops = 4.0 * 64 * float((lens.astype(np.float64) ** 2).sum()) * H
    blocks = int(np.ceil(lens / 128).sum()) * H
    print(f"{name:18s} T={T:7d}  {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  {blocks:6d} workgroups  {us * 1024 / blocks:6.2f} us per workgroup-slot round", flush=True)
