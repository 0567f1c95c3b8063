import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reprover_amd import synth, _lib
from reprover_amd.retrieval.model import PremiseRetriever
lib = _lib.load()
cfg = synth.t5_config("byt5-small"); cfg["num_layers"] = 1
sd = synth.synth_state_dict(cfg)
model = PremiseRetriever.from_state_dict(cfg, sd, 2048, "cuda:0", dtype=torch.float32)
rng = np.random.default_rng(0)
texts = [synth.synth_text(rng, n) for n in (8, 17, 33, 64, 100, 128, 180, 256, 300, 400)]
D, NP = 1472, 23
def internals(txts):
    ids, cu = model.tokenizer.packed(txts, 2048)
    out = model.encode_texts(txts); torch.cuda.synchronize()
    T = int(cu[-1]); Tp = (T + 255) // 256 * 256
    ws = model.encoder._ws
    def al(n): return (n + 255) // 256 * 256
    o = 0
    x = ws[o:o + Tp * D * 4].view(torch.float32).view(Tp, D).clone(); o += al(Tp * D * 4)
    xb = ws[o:o + Tp * D * 2].view(torch.bfloat16).view(Tp, D).clone(); o += al(Tp * D * 2)
    ssp = ws[o:o + Tp * NP * 4].view(torch.float32).view(Tp, NP).clone(); o += al(Tp * NP * 4)
    qkv = ws[o:o + Tp * 1152 * 2].view(torch.bfloat16).view(Tp, 1152).clone(); o += al(Tp * 1152 * 2)
    att = ws[o:o + Tp * 384 * 2].view(torch.bfloat16).view(Tp, 384).clone()
    return out, x, xb, ssp, cu, qkv, att
lib.rp_set_option(b"gemm_skinny", 0); lib.rp_set_option(b"debug_skip_ffn", 1)
for vo in (6, 11):
    lib.rp_set_option(b"gemm_variant_o", vo)
    ob, xb_, xbb, sb, cub, qb, ab = internals(texts)
    os_, xs, xbs, ss, cus, qs, as_ = internals([texts[7]])
    ob2, xb2, _, _, _, qb2, ab2 = internals(texts)
    print('  run-to-run x diff', (xb_-xb2).abs().max().item(), 'att diff', (ab.float()-ab2.float()).abs().max().item())
    a, b = int(cub[7]), int(cub[8])
    print(f"variant_o={vo}: emb diff {(ob[7]-os_[0]).abs().max().item():.2e}; x diff {(xb_[a:b]-xs[:b-a]).abs().max().item():.2e}; "
          f"xb diff {(xbb[a:b].float()-xbs[:b-a].float()).abs().max().item():.2e}; ssp diff {(sb[a:b]-ss[:b-a]).abs().max().item():.2e}")
    print('  qkv diff', (qb[a:b].float()-qs[:b-a].float()).abs().max().item(), 'att diff', (ab[a:b].float()-as_[:b-a].float()).abs().max().item(), 'att rows differing', sorted(set(torch.nonzero((ab[a:b].float()-as_[:b-a].float()).abs()>0)[:,0].tolist()))[:10])
    dx = (xb_[a:b]-xs[:b-a]).abs()
    if dx.max() > 0:
        idx = torch.nonzero(dx > 0); print("  #x elements differing:", len(idx), "first:", idx[:5].tolist(), "rows:", sorted(set(idx[:,0].tolist()))[:10])
    ds = (sb[a:b]-ss[:b-a]).abs()
    if ds.max() > 0:
        idx = torch.nonzero(ds > 0); print("  #ssp elements differing:", len(idx), idx[:8].tolist())
