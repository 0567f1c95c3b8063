"""A/B of TWO BUILDS of the library on one box for the TRAINING step (bench.train_step_leg at the reference's batch 8, dropout
0.1): alternating processes, median step time and per-class kernel times.   python tools/train_ab.py <libA.so> <libB.so> [rounds]
(tools/lib_ab.py is the same for the inference encode pass.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) >= 3 and sys.argv[1] != "--child":
    libs, rounds = sys.argv[1:3], int(sys.argv[3]) if len(sys.argv) > 3 else 3
    res = {l: [] for l in libs}
    for r in range(rounds):
        for l in libs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", l], capture_output=True, text=True, timeout=900)
            line = [x for x in out.stdout.splitlines() if x.startswith("{")]
            if not line:
                print(out.stdout[-2000:], out.stderr[-2000:])
                sys.exit(1)
            res[l].append(json.loads(line[-1]))
    for l in libs:
        ms = sorted(x["ms_per_step"] for x in res[l])
        ks = res[l][0]["kernel_ms_per_step"].keys()
        med = {k: sorted(x["kernel_ms_per_step"][k] for x in res[l])[len(res[l]) // 2] for k in ks}
        print(f"{l}: median {ms[len(ms) // 2]:.3f} ms (all: {' '.join(f'{m:.2f}' for m in ms)}) loss {res[l][0]['loss_first_last']} | " +
              " ".join(f"{k} {v:.3f}" for k, v in med.items() if v > 0.01))
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch

import bench
from reprover_amd import _lib, synth

_lib.LIB_PATH = os.path.abspath(sys.argv[2])
_lib.load()
dev = torch.device("cuda", 0)
cfg = synth.t5_config("byt5-small")
sd = bench.random_init_state_dict(cfg, dev, seed=synth.SEED)
print(json.dumps(bench.train_step_leg(cfg, sd, dev, int(os.environ.get("BATCH", "8")), dropout_rate=float(os.environ.get("DROPOUT", "0.1")))))
