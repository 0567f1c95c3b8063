"""Per-kernel-class time of one encoder pass as a function of the pass size (states of the benchmark's length mix):
where a 64-state pass (the reference's eval batch) loses against the 256-state pass of the headline step."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); _argv, sys.argv = sys.argv, ["x"]; spec.loader.exec_module(bench); sys.argv = _argv
from reprover_amd import _lib, synth
from reprover_amd.encoder import HipT5Encoder
from reprover_amd.tokenizer import ByT5Tokenizer
dev = torch.device("cuda:0")
cfg = synth.t5_config("byt5-small")
enc = HipT5Encoder(cfg, bench.random_init_state_dict(cfg, dev, 1), dev)
tok = ByT5Tokenizer()
rng = np.random.default_rng(synth.SEED + 100)
lens = synth.synth_lengths(rng, 256, "mix", lo=16, hi=2048)
texts = [synth.synth_state(rng, int(n) - 1) for n in lens]
for B in [int(a) for a in sys.argv[1:]] or [32, 64, 128, 256]:
    tot = {}
    ntok = 0
    for i in range(0, 256, B):
        ids_p, cu = tok.packed(texts[i:i + B], 2048)
        ntok += int(cu[-1])
        for _ in range(2):
            enc.encode_packed(ids_p, cu)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(5):
            enc.encode_packed(ids_p, cu)
        torch.cuda.synchronize()
        prof = _lib.profile_read()
        _lib.profile_enable(False)
        for k, (ms, n) in prof.items():
            if n:
                tot[k] = tot.get(k, 0.0) + ms / 5
    s = sum(tot.values())
    print(f"{256 // B} passes of {B} states ({ntok} tokens): {s:.2f} ms  " + "  ".join(f"{k} {v:.2f}" for k, v in tot.items()), flush=True)
