"""Thin callers of the C ABI used by the GPU parity tests (raw pointers in, torch tensors as
containers) + comparison helpers."""
import numpy as np
import torch

from reprover_amd import _lib


def dev():
    return torch.device("cuda:0")


def split_planes(x):
    """fp32 [M, N] -> the residual stream's two bf16 planes [2, M, N]: hi = bf16(x), lo = bf16(x - hi)."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo]).contiguous()


def merge_planes(planes):
    return planes[0].float() + planes[1].float()


def split_x24(x):
    """fp32 [M, N] -> the inference pass's 24-bit form of the residual stream (rp_encoder_kernels.h, x24_update2): the
    fp32 word rounded to its top 24 bits (half away from zero), as (hi bf16 [M, N] = that word rounded to 16 bits, half
    away from zero; ext uint8 [M, N] = the signed remainder in units of 2^-8 ulp(hi) stored BIASED by 128 = bits 8..15 of
    (the rounded word + 0x8000): round 6).  Returned as ONE uint8 buffer [M * N * 3] laid out [hi plane | ext plane] (what
    RP_EPI_RESID8 takes) plus the two views."""
    bits = x.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    q = (bits + 0x8080) & 0xFFFFFFFF
    ext = ((q >> 8) & 0xFF).to(torch.uint8)
    hi16 = (q >> 16).to(torch.int32).to(torch.int16)  # (wraps like uint16)
    M, N = x.shape
    buf = torch.empty(M * N * 3, dtype=torch.uint8, device=x.device)
    buf[: M * N * 2].view(torch.int16).view(M, N).copy_(hi16)
    buf[M * N * 2 :].view(M, N).copy_(ext)
    return buf, buf[: M * N * 2].view(torch.bfloat16).view(M, N), buf[M * N * 2 :].view(M, N)


def merge_x24(hi, ext):
    """x = float(((hi << 16) | (ext << 8)) - 0x8000) as 32-bit words."""
    w = (hi.contiguous().view(torch.int16).to(torch.int64) & 0xFFFF) << 16
    w = (w + (ext.to(torch.int64) << 8) - 0x8000) & 0xFFFFFFFF
    w = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)
    return w.view(torch.float32)


def gemm(A, W, n_valid, epilogue, out):
    """``out``: the epilogue's output; for RP_EPI_RESID an fp32 [M, n_valid] matrix that is updated in place THROUGH
    the engine's two-plane form of the residual stream (split before the call, merged after it)."""
    lib = _lib.load()
    M, K = A.shape
    N = W.shape[0]
    target = split_planes(out) if epilogue == _lib.RP_EPI_RESID else out
    _lib.check(lib.rp_dbg_gemm(_lib.ptr(A), _lib.ptr(W), _lib.ptr(target), M, N, K, n_valid, epilogue,
                               _lib.current_stream()), "rp_dbg_gemm")
    torch.cuda.synchronize()
    if epilogue == _lib.RP_EPI_RESID:
        out.copy_(merge_planes(target))
    return out


def rowscale(ssp, inv_d, eps=1e-6):
    """rs [rows] from slot-major partial sums of squares ssp [np, rows] (rowscale_kernel, as the encoder runs it)."""
    lib = _lib.load()
    np_, rows = ssp.shape
    rs = torch.empty(rows, dtype=torch.float32, device=ssp.device)
    _lib.check(lib.rp_dbg_rowscale(_lib.ptr(ssp), _lib.ptr(rs), rows, np_, inv_d, eps, _lib.current_stream()),
               "rp_dbg_rowscale")
    torch.cuda.synchronize()
    return rs


def attention(qkv, cu, tab, H):
    lib = _lib.load()
    T = qkv.shape[0]
    out = torch.zeros((T, H * 64), dtype=torch.bfloat16, device=qkv.device)
    lens = (cu[1:] - cu[:-1]).cpu()
    _lib.check(lib.rp_dbg_attention(_lib.ptr(qkv), _lib.ptr(cu), _lib.ptr(tab), _lib.ptr(out), len(lens),
                                    int(lens.max()), H, T, _lib.current_stream()), "rp_dbg_attention")
    torch.cuda.synchronize()
    return out


def sim_topk(Q, E, k, masks=None, id_offset=0, flags=0, N=None, retry_dense=True):
    """masks = (file_of i32 [N], end_key i64 [N], bits_t u32-as-i32 [F, W], own i32 [B], qk i64 [B]) device tensors.
    With flags & RP_TOPK_E_BLOCKED, E is the flat blocked copy and N must be given."""
    lib = _lib.load()
    B, D = Q.shape
    N = E.shape[0] if N is None else N
    out_s = torch.empty((B, k), dtype=torch.float32, device=Q.device)
    out_i = torch.empty((B, k), dtype=torch.int32, device=Q.device)
    out_c = torch.empty((B,), dtype=torch.int32, device=Q.device)
    nbytes = lib.rp_sim_topk_workspace_bytes(B, N, D, k, flags)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=Q.device)
    if masks is None:
        f = ek = bt = own = qk = None
        F = 0
    else:
        f, ek, bt, own, qk = masks
        F = bt.shape[0]
    _lib.check(lib.rp_sim_topk(_lib.ptr(Q), _lib.ptr(E), B, N, D, _lib.ptr(f), _lib.ptr(ek), _lib.ptr(bt), F,
                               _lib.ptr(own), _lib.ptr(qk), id_offset, k, flags, _lib.ptr(out_s), _lib.ptr(out_i),
                               _lib.ptr(out_c), _lib.ptr(ws), nbytes, _lib.current_stream()), "rp_sim_topk")
    torch.cuda.synchronize()
    if retry_dense and bool((out_c < 0).any()):  # the ABI's overflow contract, as every product caller honours it
        return sim_topk(Q, E, k, masks, id_offset, flags | _lib.RP_TOPK_DENSE, N, retry_dense=False)
    return out_i, out_s, out_c


def quantize_e4m3(X):
    """(codes uint8 [R, D], scale f32 [R]) device tensors via rp_quantize_rows_e4m3."""
    lib = _lib.load()
    X = X.contiguous()
    R, D = X.shape
    codes = torch.empty((R, D), dtype=torch.uint8, device=X.device)
    scale = torch.empty((R,), dtype=torch.float32, device=X.device)
    dt = _lib.RP_DT_F32 if X.dtype == torch.float32 else _lib.RP_DT_BF16
    _lib.check(lib.rp_quantize_rows_e4m3(_lib.ptr(X), dt, R, D, _lib.ptr(codes), _lib.ptr(scale),
                                         _lib.current_stream()), "rp_quantize_rows_e4m3")
    torch.cuda.synchronize()
    return codes, scale


def sim_topk_fp8(Q8, qs, E8, es, k, masks=None, id_offset=0, flags=0, N=None, retry_dense=True):
    """rp_sim_topk_fp8 on e4m3 codes (uint8) + per-row scales; masks as in sim_topk."""
    lib = _lib.load()
    B, D = Q8.shape
    N = E8.shape[0] if N is None else N
    out_s = torch.empty((B, k), dtype=torch.float32, device=Q8.device)
    out_i = torch.empty((B, k), dtype=torch.int32, device=Q8.device)
    out_c = torch.empty((B,), dtype=torch.int32, device=Q8.device)
    nbytes = lib.rp_sim_topk_workspace_bytes(B, N, D, k, flags)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=Q8.device)
    if masks is None:
        f = ek = bt = own = qk = None
        F = 0
    else:
        f, ek, bt, own, qk = masks
        F = bt.shape[0]
    _lib.check(lib.rp_sim_topk_fp8(_lib.ptr(Q8), _lib.ptr(qs), _lib.ptr(E8), _lib.ptr(es), B, N, D, _lib.ptr(f),
                                   _lib.ptr(ek), _lib.ptr(bt), F, _lib.ptr(own), _lib.ptr(qk), id_offset, k, flags,
                                   _lib.ptr(out_s), _lib.ptr(out_i), _lib.ptr(out_c), _lib.ptr(ws), nbytes,
                                   _lib.current_stream()), "rp_sim_topk_fp8")
    torch.cuda.synchronize()
    if retry_dense and bool((out_c < 0).any()):
        return sim_topk_fp8(Q8, qs, E8, es, k, masks, id_offset, flags | _lib.RP_TOPK_DENSE, N, retry_dense=False)
    return out_i, out_s, out_c


def topk_merge(scores, ids, counts):
    lib = _lib.load()
    R, B, k = scores.shape
    out_s = torch.empty((B, k), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((B, k), dtype=torch.int32, device=scores.device)
    out_c = torch.empty((B,), dtype=torch.int32, device=scores.device)
    nbytes = lib.rp_topk_merge_workspace_bytes(R, B, k)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=scores.device)
    _lib.check(lib.rp_topk_merge(_lib.ptr(scores), _lib.ptr(ids), _lib.ptr(counts), R, B, k, _lib.ptr(out_s),
                                 _lib.ptr(out_i), _lib.ptr(out_c), _lib.ptr(ws), nbytes, _lib.current_stream()),
               "rp_topk_merge")
    torch.cuda.synchronize()
    return out_i, out_s, out_c


from reprover_amd.synth import synth_masks  # noqa: E402,F401  (moved: bench.py --config c5 draws the same operands)


def masks_to_device(m, device):
    f, ek, bt, own, qk = m
    return (torch.from_numpy(f).to(device), torch.from_numpy(ek).to(device),
            torch.from_numpy(bt.view(np.int32)).to(device), torch.from_numpy(own).to(device),
            torch.from_numpy(qk).to(device))


def check_topk_against_scores(ids, scores, counts, S, acc, k, tol):
    """Size-independent properties of an exact masked top-k, given the full fp32 score matrix S
    [B, N] (numpy) and the accessibility predicate acc [B, N]:
    sortedness, accessibility, no duplicates, count = min(k, #accessible), reported score = S at the
    reported id (within tol), and r-th reported score within tol of the true r-th best."""
    B = S.shape[0]
    masked = np.where(acc, S, -np.inf)
    want_sorted = -np.sort(-masked, axis=1)[:, :k]
    n_acc = acc.sum(1)
    assert np.array_equal(counts, np.minimum(k, n_acc)), (counts[:8], n_acc[:8])
    for j in range(B):
        c = int(counts[j])
        row_i, row_s = ids[j, :c], scores[j, :c]
        assert np.all(np.diff(row_s) <= 0), f"row {j} not sorted"
        assert len(set(row_i.tolist())) == c, f"row {j} has duplicate ids"
        assert acc[j, row_i].all(), f"row {j} returned an inaccessible premise"
        assert np.abs(S[j, row_i] - row_s).max(initial=0) <= tol, f"row {j} score mismatch"
        assert np.abs(want_sorted[j, :c] - row_s).max(initial=0) <= tol, f"row {j} not the top-k"
        assert np.all(ids[j, c:] == -1) and np.all(np.isneginf(scores[j, c:]))
        ties = np.flatnonzero(np.diff(row_s) == 0)  # equal scores must come lower-id first
        assert np.all(row_i[ties] < row_i[ties + 1]), f"row {j} tie order"


from oracle.parity_margins import gap_rule_ids  # noqa: E402,F401  (one definition, shared with smoke())
