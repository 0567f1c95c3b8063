"""Print the measured error of every training kernel and of the end-to-end gradients (fixture G11) - no asserts: one
GPU call shows where every piece stands.  python tools/train_diag.py [kernels] [e2e]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import numpy as np
import torch

import train_helpers as th


def attempt(name, fn):
    try:
        print(f"{name}: {fn()}", flush=True)
    except Exception:
        print(f"{name}: EXCEPTION\n{traceback.format_exc()}", flush=True)


def kernels():
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3407)
    attempt("wgrad structured", th.check_wgrad_structured)
    for a in [(256, 128, 128, 1), (512, 1152, 1472, 3), (256, 200, 72, 2)]:
        attempt(f"wgrad {a}", lambda: th.check_wgrad(gen, *a))
    for a in [(256, 256, 128, 0), (256, 256, 128, 26), (512, 3584, 1472, 26)]:
        attempt(f"geglu_bwd {a}", lambda: th.check_geglu_bwd(gen, *a))
    for a in [(256, 128, 384, 0), (256, 1472, 1152, 26)]:
        attempt(f"rms_bwd_resid {a}", lambda: th.check_rms_bwd_resid(gen, *a))
    for a in [([5, 64, 129, 300, 77], 2), ([1, 2, 3], 6), ([600, 40], 2)]:
        attempt(f"attention_bwd {a}", lambda: th.check_attention_bwd(gen, *a))


def e2e():
    from reprover_amd.train import HipT5Trainer

    cfg, sd, groups, label, g = th.g11_batch(os.path.join(ROOT, "tests", "golden"))
    tr = HipT5Trainer(cfg, sd, "cuda:0", lr=float(g["lr"]), warmup_steps=int(g["warmup_steps"]))
    loss, sim = tr.contrastive_step(groups, label)
    torch.cuda.synchronize()
    print(f"e2e loss {float(loss):.6f} reference {float(g['loss']):.6f}")
    for key, (err, mx, rel) in th.grad_errors(tr, g).items():
        print(f"  {key:72s} max|d| {err:.3e}  max|ref| {mx:.3e}  rel-L2 {rel:.3e}")


if __name__ == "__main__":
    which = sys.argv[1:] or ["kernels", "e2e"]
    if "kernels" in which:
        kernels()
    if "e2e" in which:
        attempt("e2e", e2e)
