"""What the padded entry point (rp_encode_padded: grids sized for B x L, live tiles known on the device only) costs
against the packed one (rp_encode_varlen: exact grids) on the same 64 states of the benchmark's length mix."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); sys.argv = ["x"]; spec.loader.exec_module(bench)
from reprover_amd import synth
from reprover_amd.encoder import HipT5Encoder
from reprover_amd.tokenizer import ByT5Tokenizer
dev = torch.device("cuda:0")
cfg = synth.t5_config("byt5-small")
enc = HipT5Encoder(cfg, bench.random_init_state_dict(cfg, dev, 1), dev)
rng = np.random.default_rng(3)
tok = ByT5Tokenizer()
for B in (64, 256):
    lens = synth.synth_lengths(rng, B, "mix", lo=16, hi=1024)
    texts = [synth.synth_state(rng, int(n)) for n in lens]
    t = tok(texts, padding="longest", max_length=1024, truncation=True, return_tensors="pt")
    ids_d, mask_d = t.input_ids.to(dev), t.attention_mask.to(dev)
    ids_p, cu = tok.packed(texts, 1024)
    def timeit(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    a = timeit(lambda: enc.encode_padded(ids_d, mask_d, defer_check=True))
    enc.raise_pending()
    b = timeit(lambda: enc.encode_packed(ids_p, cu))
    print(f"B={B}: {int(cu[-1])} tokens, padded to {ids_d.shape[1]} ({B * ids_d.shape[1]} rows bound): padded entry {a:.3f} ms, "
          f"packed entry {b:.3f} ms (incl. its H2D of ids)", flush=True)
