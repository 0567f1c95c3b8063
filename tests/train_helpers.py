"""Callers + fp32 restatements for the training kernels (tests/test_train_kernels_gpu.py, tools/train_diag.py).
Every check returns a dict of measured errors; the tests assert on them, the diagnostic tool prints them."""
import math
import os

import numpy as np
import torch

from reprover_amd import _lib


def rand_bf16(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen, device="cuda") * scale).to(torch.bfloat16)


# ---------------------------------------------------------------------------------------------------------
def wgrad(Y, X, splits):
    lib = _lib.load()
    T, ny = Y.shape
    nx = X.shape[1]
    out = torch.full((abs(splits), ny, nx), float("nan"), dtype=torch.float32, device=Y.device)
    _lib.check(lib.rp_dbg_wgrad(_lib.ptr(Y), _lib.ptr(X), _lib.ptr(out), T, ny, nx, splits, _lib.current_stream()),
               "rp_dbg_wgrad")
    torch.cuda.synchronize()
    return out


def check_wgrad(gen, T, ny, nx, splits):
    Y, X = rand_bf16(gen, T, ny), rand_bf16(gen, T, nx)
    out = wgrad(Y, X, splits)
    ref = Y.float().T @ X.float()
    nk = T // 64
    worst_split = 0.0
    splits = abs(splits)  # (the sign selects the tile configuration)
    for s in range(splits):
        r0, r1 = (s * nk // splits) * 64, ((s + 1) * nk // splits) * 64
        part = Y[r0:r1].float().T @ X[r0:r1].float()
        worst_split = max(worst_split, (out[s] - part).abs().max().item())
    return {"max_err": (out.sum(0) - ref).abs().max().item(), "ref_max": ref.abs().max().item(),
            "worst_split_err": worst_split, "nan": int(torch.isnan(out).sum().item())}


def check_wgrad_pair(gen, T, shapes, splits):
    """Two products in one launch (rp_dbg_wgrad_pair) against the same products launched one by one: the same bits per split."""
    lib = _lib.load()
    (ny0, nx0), (ny1, nx1) = shapes
    Y0, X0, Y1, X1 = rand_bf16(gen, T, ny0), rand_bf16(gen, T, nx0), rand_bf16(gen, T, ny1), rand_bf16(gen, T, nx1)
    out0 = torch.full((splits, ny0, nx0), float("nan"), dtype=torch.float32, device="cuda")
    out1 = torch.full((splits, ny1, nx1), float("nan"), dtype=torch.float32, device="cuda")
    _lib.check(lib.rp_dbg_wgrad_pair(_lib.ptr(Y0), _lib.ptr(X0), _lib.ptr(out0), ny0, nx0, _lib.ptr(Y1), _lib.ptr(X1),
                                     _lib.ptr(out1), ny1, nx1, T, splits, _lib.current_stream()), "rp_dbg_wgrad_pair")
    torch.cuda.synchronize()
    one0, one1 = wgrad(Y0, X0, splits), wgrad(Y1, X1, splits)
    ref0, ref1 = Y0.float().T @ X0.float(), Y1.float().T @ X1.float()
    return {"same_bits": bool(torch.equal(out0.view(torch.int32), one0.view(torch.int32)) and
                              torch.equal(out1.view(torch.int32), one1.view(torch.int32))),
            "max_err": max((out0.sum(0) - ref0).abs().max().item(), (out1.sum(0) - ref1).abs().max().item()),
            "ref_max": max(ref0.abs().max().item(), ref1.abs().max().item())}


def check_wgrad_structured():
    """Operands that make any token / feature permutation visible: Y[t, o] = (t % 5 == o % 5), X[t, c] = t % 7 + c % 3."""
    T, ny, nx = 256, 256, 128
    t = torch.arange(T, device="cuda")[:, None]
    Y = ((t % 5) == (torch.arange(ny, device="cuda")[None, :] % 5)).float()
    X = ((t % 7) + (torch.arange(nx, device="cuda")[None, :] % 3)).float()
    out = wgrad(Y.to(torch.bfloat16), X.to(torch.bfloat16), 1)[0]
    return {"exact": bool(torch.equal(out, Y.T @ X))}


# ---------------------------------------------------------------------------------------------------------
def gelu_new(u):
    return 0.5 * u * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (u + 0.044715 * torch.pow(u, 3.0))))


def packed_cols(F, device):
    f = torch.arange(F, device=device)
    gate = (f // 32) * 64 + f % 32
    return gate, gate + 32


def check_geglu_bwd(gen, M, F, K, variant):
    lib = _lib.load()
    dx, W = rand_bf16(gen, M, K), rand_bf16(gen, F, K, scale=K ** -0.5)
    gu = rand_bf16(gen, M, 2 * F, scale=1.5)
    rs = torch.rand(M, generator=gen, device="cuda") + 0.5
    dzs = torch.full((M, 2 * F), float("nan"), dtype=torch.bfloat16, device="cuda")
    dots = torch.full((M,), float("nan"), dtype=torch.float32, device="cuda")
    _lib.check(lib.rp_dbg_dgrad(_lib.ptr(dx), _lib.ptr(W), M, F, K, 0, _lib.ptr(gu), _lib.ptr(rs), _lib.ptr(dzs),
                                _lib.ptr(dots), variant, _lib.current_stream()), "rp_dbg_dgrad(0)")
    torch.cuda.synchronize()
    gc, uc = packed_cols(F, "cuda")
    g = gu[:, gc].float().requires_grad_(True)
    u = gu[:, uc].float().requires_grad_(True)
    dff = dx.float() @ W.float().T
    (gelu_new(g) * u * dff).sum().backward()
    ref = torch.zeros(M, 2 * F, device="cuda")
    ref[:, gc] = g.grad * rs[:, None]
    ref[:, uc] = u.grad * rs[:, None]
    ref_dot = (g.grad * g.detach() + u.grad * u.detach()).sum(1)
    return {"dzs_err": (dzs.float() - ref).abs().max().item(), "dzs_max": ref.abs().max().item(),
            "dot_err": (dots - ref_dot).abs().max().item(), "dot_max": ref_dot.abs().max().item(),
            "nan": int(torch.isnan(dzs.float()).sum().item() + torch.isnan(dots).sum().item())}


def geglu_bwd_outputs(gen_seed, M, F, K, variant, mixed):
    """dzs / row dots of one gated-GELU backward GEMM with the mixed launch (full + half tiles) switched on or off."""
    lib = _lib.load()
    gen = torch.Generator(device="cuda").manual_seed(gen_seed)
    dx, W = rand_bf16(gen, M, K), rand_bf16(gen, F, K, scale=K ** -0.5)
    gu = rand_bf16(gen, M, 2 * F, scale=1.5)
    rs = torch.rand(M, generator=gen, device="cuda") + 0.5
    dzs = torch.full((M, 2 * F), float("nan"), dtype=torch.bfloat16, device="cuda")
    dots = torch.full((M,), float("nan"), dtype=torch.float32, device="cuda")
    _lib.check(lib.rp_set_option(b"gemm_mixed_bwd", int(mixed)), "opt")
    try:
        _lib.check(lib.rp_dbg_dgrad(_lib.ptr(dx), _lib.ptr(W), M, F, K, 0, _lib.ptr(gu), _lib.ptr(rs), _lib.ptr(dzs),
                                    _lib.ptr(dots), variant, _lib.current_stream()), "rp_dbg_dgrad(0)")
        torch.cuda.synchronize()
    finally:
        _lib.check(lib.rp_set_option(b"gemm_mixed_bwd", 1), "opt")
    return dzs, dots


def check_rms_bwd_resid(gen, M, N, K, variant):
    import hip_helpers as hh

    lib = _lib.load()
    A, W = rand_bf16(gen, M, K), rand_bf16(gen, N, K, scale=K ** -0.5)
    x = rand_bf16(gen, M, N)
    rc = torch.randn(M, generator=gen, device="cuda") * 0.3
    dx0 = torch.randn(M, N, generator=gen, device="cuda")
    planes = hh.split_planes(dx0)
    start = hh.merge_planes(planes)
    _lib.check(lib.rp_dbg_dgrad(_lib.ptr(A), _lib.ptr(W), M, N, K, 1, _lib.ptr(x), _lib.ptr(rc), _lib.ptr(planes), None,
                                variant, _lib.current_stream()), "rp_dbg_dgrad(1)")
    torch.cuda.synchronize()
    got = hh.merge_planes(planes)
    ref = start + A.float() @ W.float().T - x.float() * rc[:, None]
    # hi = bf16(x) up to the re-split's own rounding: never further than one bf16 ulp from the fp32 sum
    return {"err": (got - ref).abs().max().item(), "ref_max": ref.abs().max().item(),
            "hi_err": (planes[0].float() - ref).abs().max().item()}


# ---------------------------------------------------------------------------------------------------------
def attention_bwd(qkv, cu, tab, H, datt, rows_total):
    lib = _lib.load()
    T = qkv.shape[0]
    lse = torch.zeros((H, rows_total), dtype=torch.float32, device=qkv.device)
    att = torch.zeros((T, H * 64), dtype=torch.bfloat16, device=qkv.device)
    dqkv = torch.zeros((T, 3 * H * 64), dtype=torch.bfloat16, device=qkv.device)
    dtab = torch.zeros((257, H), dtype=torch.float32, device=qkv.device)
    _lib.check(lib.rp_dbg_attention_bwd(_lib.ptr(qkv), None, _lib.ptr(datt), _lib.ptr(cu), _lib.ptr(tab), len(cu) - 1, H,
                                        rows_total, _lib.ptr(lse), _lib.ptr(att), _lib.ptr(dqkv), _lib.ptr(dtab),
                                        _lib.current_stream()), "rp_dbg_attention_bwd")
    torch.cuda.synchronize()
    return att, lse, dqkv, dtab


def check_attention_bwd(gen, lens, H):
    maxd = 128
    T = int(sum(lens))
    Tp = (T + 255) // 256 * 256
    inner = H * 64
    qkv = torch.zeros((Tp, 3 * inner), dtype=torch.bfloat16, device="cuda")
    qkv[:T] = rand_bf16(gen, T, 3 * inner, scale=0.5)
    datt = torch.zeros((Tp, inner), dtype=torch.bfloat16, device="cuda")
    datt[:T] = rand_bf16(gen, T, inner)
    tab = torch.randn(H, 2 * maxd + 1, generator=gen, device="cuda")
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device="cuda")
    att, lse, dqkv, dtab = attention_bwd(qkv, cu, tab, H, datt, Tp)
    # fp32 restatement with autograd, sequence by sequence
    tab_r = tab.clone().requires_grad_(True)
    ref_dqkv = torch.zeros(T, 3 * inner, device="cuda")
    ref_att = torch.zeros(T, inner, device="cuda")
    ref_lse = torch.zeros(H, T, device="cuda")
    s0 = 0
    for L in lens:
        x = qkv[s0 : s0 + L].float().clone().requires_grad_(True)
        q, k, v = [x[:, i * inner : (i + 1) * inner].view(L, H, 64).transpose(0, 1) for i in range(3)]
        rel = (torch.arange(L, device="cuda")[None, :] - torch.arange(L, device="cuda")[:, None]).clamp(-maxd, maxd) + maxd
        s = q @ k.transpose(1, 2) + tab_r[:, rel]
        p = torch.softmax(s, dim=-1)
        o = (p @ v).transpose(0, 1).reshape(L, inner)
        (o * datt[s0 : s0 + L].float()).sum().backward()
        ref_dqkv[s0 : s0 + L] = x.grad
        ref_att[s0 : s0 + L] = o.detach()
        ref_lse[:, s0 : s0 + L] = torch.logsumexp(s.detach(), dim=-1) * math.log2(math.e)
        s0 += L
    ref_dtab = tab_r.grad.T  # [257, H]
    out = {}
    for name, a, b in (("dq", dqkv[:T, :inner], ref_dqkv[:, :inner]), ("dk", dqkv[:T, inner : 2 * inner], ref_dqkv[:, inner : 2 * inner]),
                       ("dv", dqkv[:T, 2 * inner :], ref_dqkv[:, 2 * inner :]), ("att", att[:T], ref_att),
                       ("lse", lse[:, :T], ref_lse), ("dtab", dtab, ref_dtab)):
        out[name + "_err"] = (a.float() - b).abs().max().item()
        out[name + "_max"] = b.abs().max().item()
    out["nan"] = int(torch.isnan(dqkv.float()).sum().item() + torch.isnan(dtab).sum().item())
    return out


# ---------------------------------------------------------------------------------------------------------
def g11_batch(golden_dir):
    """The tiny training batch of fixture G11 as the reference's collate produced it: (cfg, state dict, groups, label, g)."""
    from reprover_amd import synth
    from reprover_amd.tokenizer import ByT5Tokenizer

    g = np.load(os.path.join(golden_dir, "g11_train_backward.npz"), allow_pickle=True)
    return (synth.t5_config("tiny"),) + _batch_from_texts(g, synth.t5_config("tiny"), ByT5Tokenizer())


def _batch_from_texts(g, cfg, tok):
    from reprover_amd import synth

    sd = synth.synth_state_dict(cfg, seed=int(g["weight_seed"]))
    L = int(g["max_seq_len"])

    def enc(texts):
        b = tok(list(texts), padding="longest", max_length=L, truncation=True, return_tensors="pt")
        return b.input_ids, b.attention_mask

    groups = [enc(g["context_texts"]), enc(g["pos_texts"])] + [enc(t) for t in g["neg_texts"]]
    return sd, groups, torch.from_numpy(g["label"]), g


def grad_errors(trainer, golden, prefix="grad/"):
    """{HF key: (max |Δ|, max |ref|, relative L2 error)} of the engine's gradients against a fixture's."""
    out = {}
    for key, gv in trainer.named_gradients():
        ref = torch.from_numpy(golden[prefix + key]).to(gv.device)
        d = (gv - ref)
        out[key] = (d.abs().max().item(), ref.abs().max().item(), (d.norm() / (ref.norm() + 1e-30)).item())
    return out
