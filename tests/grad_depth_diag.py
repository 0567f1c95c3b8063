"""Per-tensor gradient error of the engine against the oracle's fp32 autograd at full depth, layer by layer
(python tests/grad_depth_diag.py [num_layers]): how does the bf16-operand backward's error grow towards the input?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from oracle import train_ref
from reprover_amd import synth
from reprover_amd.tokenizer import ByT5Tokenizer
from reprover_amd.train import HipT5Trainer

L = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cfg = synth.t5_config("byt5-small"); cfg["num_layers"] = L
sd = synth.synth_state_dict(cfg, seed=21)
rng = np.random.default_rng(22)
ctx = [synth.synth_state(rng, int(n)) for n in (90, 260)]
pos = [synth.synth_text(rng, int(n)) for n in (70, 150)]
neg = [[synth.synth_text(rng, int(n)) for n in (40, 200)]]
label = np.array([[1.0, 0.0, 0.0, 1.0], [0.0, 1.0, 0.0, 0.0]], dtype=np.float32)
loss_ref, grads_ref = train_ref.forward_backward(cfg, sd, ctx, pos, neg, label, 512)
# the same backward with every weight rounded to bf16 first: how much of the difference is the weights' rounding alone?
sd_bf = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
loss_bf, grads_bf = train_ref.forward_backward(cfg, sd_bf, ctx, pos, neg, label, 512)
tok = ByT5Tokenizer()
enc = lambda texts: (lambda b: (b.input_ids, b.attention_mask))(tok(list(texts), padding="longest", max_length=512, truncation=True, return_tensors="pt"))
tr = HipT5Trainer(cfg, sd, "cuda:0", lr=1e-4)
loss, _ = tr.contrastive_step([enc(ctx), enc(pos)] + [enc(n) for n in neg], torch.from_numpy(label))
print(f"loss engine {float(loss):.6f} oracle {loss_ref:.6f} oracle(bf16 weights) {loss_bf:.6f}")
for key, gv in tr.named_gradients():
    ref = torch.from_numpy(grads_ref[key]); bf = torch.from_numpy(grads_bf[key]); g = gv.cpu()
    rel = lambda a: ((a - ref).norm() / (ref.norm() + 1e-30)).item()
    cosv = torch.nn.functional.cosine_similarity(g.reshape(1, -1).double(), ref.reshape(1, -1).double()).item()
    rel_bf = ((g - bf).norm() / (bf.norm() + 1e-30)).item()
    print(f"{key:70s} engine rel-L2 {rel(g):.3e} cos {cosv:.5f} | fp32 oracle on bf16-rounded weights rel-L2 {rel(bf):.3e} | engine vs THAT {rel_bf:.3e} | norm {ref.norm().item():.3e}")
