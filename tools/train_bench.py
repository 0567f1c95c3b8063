"""The training-step leg of bench.py on its own (python tools/train_bench.py [batch sizes...]): ms per step, per-class
kernel times, MFMA fractions of the backward GEMMs."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

import bench
from reprover_amd import synth

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    cfg = synth.t5_config("byt5-small")
    sd = bench.random_init_state_dict(cfg, dev, seed=synth.SEED)
    p = float(os.environ.get("DROPOUT", "0.1"))
    if os.environ.get("TRAIN_DBG"):  # experiment bits of an RP_EXPERIMENTS=1 build (rp_set_option "train_dbg")
        from reprover_amd import _lib

        _lib.check(_lib.load().rp_set_option(b"train_dbg", int(os.environ["TRAIN_DBG"])), "rp_set_option")
    for bsz in [int(a) for a in sys.argv[1:]] or [8, 64]:
        print(json.dumps({f"batch{bsz}": bench.train_step_leg(cfg, sd, dev, bsz, dropout_rate=p)}), flush=True)
