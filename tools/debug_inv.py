import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reprover_amd import synth, _lib
from reprover_amd.retrieval.model import PremiseRetriever
lib = _lib.load()
cfg = synth.t5_config("byt5-small"); cfg["num_layers"] = int(os.environ.get("LAYERS", 2))
sd = synth.synth_state_dict(cfg)
model = PremiseRetriever.from_state_dict(cfg, sd, 2048, "cuda:0", dtype=torch.float32)
rng = np.random.default_rng(0)
texts = [synth.synth_text(rng, n) for n in (8, 17, 33, 64, 100, 128, 180, 256, 300, 400)]
for skinny in (1, 0):
    for vo in (11, 6):
        lib.rp_set_option(b"gemm_skinny", skinny); lib.rp_set_option(b"gemm_variant_o", vo)
        solo = torch.cat([model.encode_texts([t]) for t in texts])
        batch = model.encode_texts(texts)
        rev = model.encode_texts(texts[::-1]).flip(0)
        d1 = (solo - batch).abs().amax(1); d2 = (rev - batch).abs().amax(1)
        print(f"skinny={skinny} variant_o={vo}: solo-vs-batch max per row {[f'{x:.1e}' for x in d1.tolist()]}")
        print(f"                       rev-vs-batch  max per row {[f'{x:.1e}' for x in d2.tolist()]}")
