"""A/B of engine option sets inside the benchmark's encode pass (256 states of the length mix, ByT5-small), all in
ONE process, interleaved rounds:   python tools/step_ab.py "name:opt=v,opt=v" "name2:..." [ROUNDS=4 STEPS=4]
Prints per configuration the median ms per pass and the per-kernel split (rp_profile_*)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from reprover_amd import _lib, synth, tokenizer
from reprover_amd.encoder import HipT5Encoder

if os.environ.get("LIB"):  # another build of the library (A/B of two source states on one box: run the tool twice)
    _lib.LIB_PATH = os.path.abspath(os.environ["LIB"])
lib = _lib.load()
dev = torch.device("cuda", 0)
model = os.environ.get("MODEL", "byt5-small")
cfg = synth.t5_config(model)
sd = bench.random_init_state_dict(cfg, dev, seed=synth.SEED)
enc = HipT5Encoder(cfg, sd, dev, torch.bfloat16)
B = int(os.environ.get("B", "256"))
rng = np.random.default_rng(synth.SEED + 100)
lens = synth.synth_lengths(rng, B, "mix", lo=16, hi=2048)
txt = [synth.synth_state(rng, int(n) - 1) for n in lens]
ids_np, cu_np = tokenizer.encode_packed(txt, 2048)
T, max_len = int(cu_np[-1]), int(np.diff(cu_np).max())
ids_d, cu_d = torch.from_numpy(ids_np).to(dev), torch.from_numpy(cu_np).to(dev)
out = torch.empty((B, cfg["d_model"]), dtype=torch.bfloat16, device=dev)
print(f"{model} B={B} tokens={T}", flush=True)

DEFAULTS = {"gemm_variant_all": -1, "gemm_tail_split": 1, "gemm_group_m": 8, "gemm_persist": 9, "gemm_rs_lds": 0, "pool_chunk": 64, "gemm_edge_layout": 1, "gemm_tail_variant": 30, "gemm_mixed": 20}
confs = []
for a in sys.argv[1:]:
    name, _, rest = a.partition(":")
    opts = dict(DEFAULTS)
    for kv in filter(None, rest.split(",")):
        k, v = kv.split("=")
        opts[k] = int(v)
    confs.append((name, opts))

def apply(opts):
    for k in ("gemm_variant_all",) + tuple(x for x in opts if x != "gemm_variant_all"):  # _all first: it resets the others
        _lib.check(lib.rp_set_option(k.encode(), opts[k]), k)

def run():
    enc.encode_packed_device(ids_d, cu_d, B, T, max_len, out)

ROUNDS, STEPS = int(os.environ.get("ROUNDS", "4")), int(os.environ.get("STEPS", "4"))
ref = None
times = {n: [] for n, _ in confs}
split = {}
for rnd in range(ROUNDS + 1):
    for name, opts in confs:
        apply(opts)
        if rnd == 0:
            run(); torch.cuda.synchronize()
            o = out.float().clone()
            if ref is None:
                ref = o
            print(f"{name:28s} max |d emb| vs first configuration {float((o - ref).abs().max()):.3e}", flush=True)
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        run()
        e0.record()
        for _ in range(STEPS):
            run()
        e1.record(); torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / STEPS)
        if rnd == ROUNDS:
            _lib.profile_enable(True)
            for _ in range(STEPS):
                run()
            torch.cuda.synchronize()
            p = _lib.profile_read()
            _lib.profile_enable(False)
            split[name] = {k: v[0] / STEPS for k, v in p.items() if v[1]}
apply(DEFAULTS)
for name, _ in confs:
    ts = sorted(times[name])
    s = split[name]
    keys = ("gemm_qkv", "gemm_o", "gemm_wi", "gemm_wo", "attention", "embed", "pool", "rmsnorm")
    print(f"{name:28s} median {ts[len(ts)//2]:7.3f} best {ts[0]:7.3f} ms | " + " ".join(f"{k[5:] if k.startswith('gemm_') else k} {s.get(k, 0):6.3f}" for k in keys), flush=True)
