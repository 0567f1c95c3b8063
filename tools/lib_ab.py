"""A/B of TWO BUILDS of the library on one box: the bench's encode pass (256 states of the length mix) under each, alternating
processes, and the embeddings compared bit for bit.   python tools/lib_ab.py <libA.so> <libB.so> [rounds]
(step_ab.py compares option sets inside one build; this compares source states.)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) >= 3 and sys.argv[1] != "--child":
    libs, rounds = sys.argv[1:3], int(sys.argv[3]) if len(sys.argv) > 3 else 3
    res = {l: [] for l in libs}
    for r in range(rounds):
        for l in libs:
            path, _, envs = l.partition("@")  # "lib.so@VAR=value,VAR2=value": the same build under another environment
            env = dict(os.environ, **dict(kv.split("=") for kv in envs.split(",") if kv))
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path, f"/tmp/lib_ab_{libs.index(l)}.pt"],
                                 capture_output=True, text=True, timeout=600, env=env)
            line = [x for x in out.stdout.splitlines() if x.startswith("{")]
            if not line:
                print(out.stdout[-2000:], out.stderr[-2000:]); sys.exit(1)
            res[l].append(json.loads(line[-1]))
    import torch
    a, b = torch.load("/tmp/lib_ab_0.pt"), torch.load("/tmp/lib_ab_1.pt")
    print("same bits:", bool(torch.equal(a, b)), " max |d|:", float((a.float() - b.float()).abs().max()))
    for l in libs:
        ms = sorted(x["ms"] for x in res[l]); ks = res[l][0]["kernels"].keys()
        med = {k: sorted(x["kernels"][k] for x in res[l])[len(res[l]) // 2] for k in ks}
        print(f"{l}: median {ms[len(ms) // 2]:.3f} ms | " + " ".join(f"{k} {v:.3f}" for k, v in med.items() if v > 0.05))
    sys.exit(0)
lib_path, out_path = sys.argv[2], sys.argv[3]
sys.path.insert(0, ROOT)
import numpy as np, torch, time
import bench
from reprover_amd import _lib, synth, tokenizer
from reprover_amd.encoder import HipT5Encoder
_lib.LIB_PATH = os.path.abspath(lib_path)
import ctypes
_probe = ctypes.CDLL(_lib.LIB_PATH)  # an OLDER build may lack debug entry points added since: bind what it has
for _name in [n for n in _lib.SIGNATURES if not hasattr(_probe, n)]:
    assert _name.startswith("rp_dbg_"), _name
    del _lib.SIGNATURES[_name]
lib = _lib.load()
dev = torch.device("cuda", 0)
cfg = synth.t5_config("byt5-small")
enc = HipT5Encoder(cfg, bench.random_init_state_dict(cfg, dev, seed=synth.SEED), dev, torch.bfloat16)
rng = np.random.default_rng(synth.SEED + 100)
lens = synth.synth_lengths(rng, 256, "mix", lo=16, hi=2048)
ids_np, cu_np = tokenizer.encode_packed([synth.synth_state(rng, int(n) - 1) for n in lens], 2048)
T, max_len = int(cu_np[-1]), int(np.diff(cu_np).max())
ids_d, cu_d = torch.from_numpy(ids_np).to(dev), torch.from_numpy(cu_np).to(dev)
out = torch.empty((256, cfg["d_model"]), dtype=torch.bfloat16, device=dev)
for _ in range(3):
    enc.encode_packed_device(ids_d, cu_d, 256, T, max_len, out)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 10
for _ in range(N):
    enc.encode_packed_device(ids_d, cu_d, 256, T, max_len, out)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / N * 1e3
_lib.profile_enable(True)
for _ in range(N):
    enc.encode_packed_device(ids_d, cu_d, 256, T, max_len, out)
torch.cuda.synchronize()
prof = _lib.profile_read(); _lib.profile_enable(False)
torch.save(out.view(torch.int16).cpu(), out_path)
print(json.dumps({"ms": ms, "kernels": {k: v[0] / N for k, v in prof.items()}}))
