"""Each HIP kernel against a plain fp32 restatement of the same op, through the C ABI."""
import math

import numpy as np
import pytest
import torch

import hip_helpers as hh
from oracle import common_ref
from reprover_amd import _lib

pytestmark = pytest.mark.gpu


def _rand_bf16(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen, device="cuda") * scale).to(torch.bfloat16)


@pytest.fixture(scope="module")
def gen():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    g = torch.Generator(device="cuda")
    g.manual_seed(3407)
    return g


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 384, 96), (384, 1472, 1472), (128, 200, 384)])
def test_gemm_store_bf16(gen, M, N, K):
    A, W = _rand_bf16(gen, M, K), _rand_bf16(gen, N, K)
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    hh.gemm(A, W, N, _lib.RP_EPI_STORE_BF16, out)
    ref = A.float() @ W.float().T
    err = (out.float() - ref).abs().max().item()
    assert err <= 2 ** -8 * ref.abs().max().item() + 1e-3, err  # one bf16 rounding of the result


def test_gemm_detects_transposes_with_structured_operands(gen):
    # A = row index pattern, W = column index pattern: any row/col swap changes the answer
    M, N, K = 128, 256, 64
    A = torch.zeros(M, K, device="cuda")
    A[:, 0] = torch.arange(M, device="cuda") % 7
    A[:, 1] = 1.0
    W = torch.zeros(N, K, device="cuda")
    W[:, 0] = 1.0
    W[:, 1] = (torch.arange(N, device="cuda") % 5) * 8.0
    out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    hh.gemm(A.to(torch.bfloat16), W.to(torch.bfloat16), N, _lib.RP_EPI_STORE_BF16, out)
    assert torch.equal(out.float(), A @ W.T)


def test_gemm_residual_two_planes(gen):
    """x += A W^T on the residual stream's two bf16 planes (hi = bf16(x), lo = bf16(x - hi)): the update must be the
    fp32 sum to 2^-17 relative (one re-split), and hi must be the bf16 rounding of the result."""
    M, N, K = 256, 1472, 384
    A, W = _rand_bf16(gen, M, K), _rand_bf16(gen, N, K, scale=K ** -0.5)
    x = torch.randn(M, N, generator=gen, device="cuda")
    planes = hh.split_planes(x)
    x0 = hh.merge_planes(planes)
    assert (x0 - x).abs().max().item() <= 2 ** -17 * x.abs().max().item()
    ref = x0 + A.float() @ W.float().T
    lib = _lib.load()
    _lib.check(lib.rp_dbg_gemm(A.data_ptr(), W.data_ptr(), planes.data_ptr(), M, N, K, N, _lib.RP_EPI_RESID,
                               _lib.current_stream()), "rp_dbg_gemm")
    torch.cuda.synchronize()
    got = hh.merge_planes(planes)
    assert (got - ref).abs().max().item() < 2e-4
    # hi is the bf16 rounding of the updated value: the stored remainder is at most half a bf16 step of hi (bit-equal
    # to bf16(hi + lo) except where lo's own rounding lands the sum exactly on a tie)
    hi = planes[0].float()
    assert ((got - hi).abs() <= 2 ** -8 * hi.abs() + 1e-30).all(), "lo must be a rounding remainder of hi"
    assert (planes[0] != got.to(torch.bfloat16)).float().mean().item() < 1e-2


def test_gemm_residual_24_bit_form(gen):
    """The inference pass's residual stream: a bf16 plane + a one-byte extension plane (stored biased by 128 since round 6),
    x = float(((hi << 16) | (ext << 8)) - 0x8000), i.e. the fp32 word of x rounded to its top 24 bits (x24_update2).  The representation is specified to the bit: the
    update must equal the host's restatement of it EXACTLY wherever the fp32 sums agree, hi must be the 24-bit word rounded
    to 16 bits (half away from zero), and the sums of squares must be those of the values as stored."""
    M, N, K = 256, 1472, 384
    np_ = (N + 63) // 64
    A, W = _rand_bf16(gen, M, K), _rand_bf16(gen, N, K, scale=K ** -0.5)
    x = torch.randn(M, N, generator=gen, device="cuda") * 3.0
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 2.0 ** -20, -(2.0 ** 20), 1.0 + 2.0 ** -9, -(1.0 + 2.0 ** -9)], device="cuda")
    buf, hi, ext = hh.split_x24(x)
    x0 = hh.merge_x24(hi, ext)
    assert (x0 - x).abs().max().item() <= 2 ** -16 * x.abs().max().item()  # 16 significant bits
    assert ((x0 - hi.float()).abs() <= 2 ** -8 * hi.float().abs() + 1e-30).all()
    delta = A.float() @ W.float().T
    lib = _lib.load()
    ssp = torch.full((np_, M), float("nan"), device="cuda")
    _lib.check(lib.rp_dbg_gemm_fused(A.data_ptr(), W.data_ptr(), buf.data_ptr(), M, N, K, N, _lib.RP_EPI_RESID8, None, 0, 0.0, 0.0,
                                     None, ssp.data_ptr(), np_, _lib.current_stream()), "fused")
    torch.cuda.synchronize()
    got = hh.merge_x24(hi, ext)
    ref = x0 + delta
    assert ((got - ref).abs() <= 2 ** -16 * ref.abs() + 2e-6).all()  # 16 significant bits (+ the MFMA's own summation order)
    # bit-level: re-encoding what the kernel stored reproduces both planes (the stored pair is canonical) ...
    _, hi2, ext2 = hh.split_x24(got)
    assert torch.equal(hi2.view(torch.int16), hi.view(torch.int16)) and torch.equal(ext2, ext)
    # ... and it is the 24-bit rounding of an fp32 sum that agrees with torch's to an ulp or two of fp32 (MFMA order)
    _, hi3, ext3 = hh.split_x24(ref)
    same = (hi3.view(torch.int16) == hi.view(torch.int16)) & (ext3 == ext)
    assert same.float().mean().item() > 0.98
    assert (hh.merge_x24(hi3, ext3) - got).abs().max().item() <= 2 ** -15 * ref.abs().max().item()
    want = (got.double() ** 2).view(M, np_, 64).sum(-1).float().T
    assert not torch.isnan(ssp).any() and (ssp - want).abs().max().item() <= 1e-4 * want.abs().max().item()
    # the plain entry point (no statistics) stores the same planes
    buf_b, hi_b, ext_b = hh.split_x24(x)
    _lib.check(lib.rp_dbg_gemm(A.data_ptr(), W.data_ptr(), buf_b.data_ptr(), M, N, K, N, _lib.RP_EPI_RESID8,
                               _lib.current_stream()), "rp_dbg_gemm")
    torch.cuda.synchronize()
    assert torch.equal(buf_b, buf)


def test_gemm_geglu(gen):
    M, F, K = 128, 256, 128
    A = _rand_bf16(gen, M, K)
    w0, w1 = _rand_bf16(gen, F, K, scale=K ** -0.5), _rand_bf16(gen, F, K, scale=K ** -0.5)
    W = torch.empty(2 * F, K, dtype=torch.bfloat16, device="cuda")  # 32 gate rows / 32 up rows interleaved
    Wv = W.view(F // 32, 2, 32, K)
    Wv[:, 0] = w0.view(F // 32, 32, K)
    Wv[:, 1] = w1.view(F // 32, 32, K)
    out = torch.empty((M, F), dtype=torch.bfloat16, device="cuda")
    hh.gemm(A, W, 2 * F, _lib.RP_EPI_GEGLU_BF16, out)
    g = A.float() @ w0.float().T
    u = A.float() @ w1.float().T
    ref = 0.5 * g * (1 + torch.tanh(math.sqrt(2 / math.pi) * (g + 0.044715 * g ** 3))) * u
    assert (out.float() - ref).abs().max().item() <= 2 ** -8 * ref.abs().max().item() + 1e-3


def test_gemm_geglu_wide_range_of_gate_values(gen):
    """gelu_new is evaluated as u * rcp(1 + exp2(-2 z log2 e)): check the whole range the exponential
    sees, from gates where it overflows to +inf (u << 0: result -0) to gates where it underflows to 0
    (u >> 0: result u), against the tanh form in fp64; relative error stays at bf16 rounding."""
    M, F, K = 128, 64, 32
    A = torch.zeros(M, K, device="cuda")
    A[:, 0] = 1.0
    w0 = torch.zeros(F, K, device="cuda")
    w1 = torch.zeros(F, K, device="cuda")
    gates = torch.linspace(-120.0, 120.0, F, device="cuda").to(torch.bfloat16).float()
    gates[F // 2] = 0.0
    w0[:, 0] = gates          # gate value of feature f = gates[f], the same for every token
    w1[:, 0] = 1.5            # up value
    W = torch.empty(2 * F, K, dtype=torch.bfloat16, device="cuda")
    Wv = W.view(F // 32, 2, 32, K)
    Wv[:, 0] = w0.to(torch.bfloat16).view(F // 32, 32, K)
    Wv[:, 1] = w1.to(torch.bfloat16).view(F // 32, 32, K)
    out = torch.full((M, F), float("nan"), dtype=torch.bfloat16, device="cuda")
    hh.gemm(A.to(torch.bfloat16), W, 2 * F, _lib.RP_EPI_GEGLU_BF16, out)
    g = gates.double()
    ref = (0.5 * g * (1 + torch.tanh(math.sqrt(2 / math.pi) * (g + 0.044715 * g ** 3))) * 1.5).float()
    got = out.float()
    assert torch.isfinite(got).all()
    assert (got - ref[None, :]).abs().max().item() <= 2 ** -8 * ref.abs().max().item()
    assert ((got - ref[None, :]).abs() <= 2 ** -7 * ref.abs()[None, :] + 1e-30).all()


@pytest.mark.parametrize("D", [128, 1472, 1536])
def test_rmsnorm_statistic_as_the_product_computes_it(gen, D):
    """T5 RMSNorm (HF:59-72) has no pass of its own in the engine: producers of x emit per-64-feature partial sums of
    squares (checked with every GEMM tile configuration in test_gemm_fused_rmsnorm_pieces_all_variants), the
    rowscale kernel turns them into rs = rsqrt(mean(x^2) + eps), consumers multiply by rs[token] (LN weight folded into
    their weights).  This pins the rowscale kernel: slot sums in index order, fp32."""
    rows = 260
    x = torch.randn(rows, D, generator=gen, device="cuda") * 3
    np_ = (D + 63) // 64
    pad = np_ * 64 - D
    ssp = torch.nn.functional.pad(x, (0, pad)).pow(2).view(rows, np_, 64).sum(-1).T.contiguous()  # slot-major [np, rows]
    rs = hh.rowscale(ssp, 1.0 / D)
    ref = torch.rsqrt(x.double().pow(2).mean(-1) + 1e-6).float()
    assert ((rs - ref).abs() <= 2e-6 * ref).all()
    w = torch.rand(D, generator=gen, device="cuda") + 0.5
    h = w * x * rs[:, None]  # what the consumers apply (w folded into the weights there)
    want = w * x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)
    assert (h - want).abs().max().item() <= 1e-5 * want.abs().max().item()


def _attention_ref(qkv, cu, tab, H):
    T = qkv.shape[0]
    inner = H * 64
    out = torch.zeros(T, inner, device=qkv.device)
    q, k, v = qkv[:, :inner].float(), qkv[:, inner : 2 * inner].float(), qkv[:, 2 * inner :].float()
    cu = cu.tolist()
    for b in range(len(cu) - 1):
        s0, s1 = cu[b], cu[b + 1]
        L = s1 - s0
        pos = torch.arange(L, device=qkv.device)
        rel = (pos[None, :] - pos[:, None]).clamp(-128, 128) + 128
        for h in range(H):
            sl = slice(h * 64, (h + 1) * 64)
            s = q[s0:s1, sl] @ k[s0:s1, sl].T + tab[h][rel]  # no 1/sqrt(d) scaling in T5
            out[s0:s1, sl] = torch.softmax(s, dim=-1) @ v[s0:s1, sl]
    return out


@pytest.mark.parametrize("lens", [[1, 5, 64, 65, 127, 128, 129, 200, 300], [700], [2048, 33],
                                  # more sequences than the work-list kernel has threads, every short-length bucket
                                  [int(x) for x in np.random.default_rng(1).integers(1, 150, 1300)]])
def test_attention(gen, lens):
    H = 2
    T = sum(lens)
    Tp = (T + 127) // 128 * 128
    qkv = _rand_bf16(gen, Tp, 3 * H * 64)
    qkv[:, : H * 64] *= 0.5  # logits std ~ 4: sharp but finite softmax
    tab = torch.randn(H, 257, generator=gen, device="cuda")
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device="cuda")
    out = hh.attention(qkv, cu, tab, H)
    ref = _attention_ref(qkv[:T], cu, tab, H)
    err = (out[:T].float() - ref).abs().max().item()
    assert err < 3e-2, err  # p and the output pass through bf16
    assert torch.count_nonzero(out[T:]) == 0, "rows past the last sequence must not be written"


def test_attention_online_softmax_rescale_is_exercised(gen):
    """Force the running-max to jump at a late key tile (cdna guide rule 26): one key far into the
    sequence dominates one query row."""
    H, L = 1, 320
    qkv = _rand_bf16(gen, 384, 3 * 64, scale=0.3)
    qkv[7, :64] = 2.0  # query 7
    qkv[300, 64:128] = 2.0  # key 300 aligned with it: logit 256 vs O(1) elsewhere
    tab = torch.zeros(H, 257, device="cuda")
    cu = torch.tensor([0, L], dtype=torch.int32, device="cuda")
    out = hh.attention(qkv, cu, tab, H)
    ref = _attention_ref(qkv[:L], cu, tab, H)
    assert (out[:L].float() - ref).abs().max().item() < 3e-2
    assert (out[7].float() - qkv[300, 128:192].float()).abs().max().item() < 1e-2  # row 7 == v[300]


def _scores(Q, E):
    return (Q.double() @ E.double().T).float().cpu().numpy()  # exact products, fp64 accumulate


@pytest.mark.parametrize("B,N,D,k,masked", [
    (1, 1000, 64, 10, True), (37, 5000, 128, 100, True), (256, 20000, 1472, 100, True),
    (130, 40000, 256, 100, True), (64, 70000, 128, 7, False), (3, 300, 32, 100, True)])
def test_sim_topk(gen, B, N, D, k, masked):
    rng = np.random.default_rng(B * 7 + N)
    E = torch.nn.functional.normalize(torch.randn(N, D, generator=gen, device="cuda"), dim=1).to(torch.bfloat16)
    Q = torch.nn.functional.normalize(torch.randn(B, D, generator=gen, device="cuda"), dim=1).to(torch.bfloat16)
    if masked:
        m, acc = hh.synth_masks(rng, N, B, F=max(2, N // 25))
        dm = hh.masks_to_device(m, Q.device)
    else:
        dm, acc = None, np.ones((B, N), dtype=bool)
    S = _scores(Q, E)
    ids, sc, cnt = hh.sim_topk(Q, E, k, dm)
    hh.check_topk_against_scores(ids.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy(), S, acc, k, tol=5e-6)
    # the dense single-pass path must give bit-identical answers
    ids2, sc2, cnt2 = hh.sim_topk(Q, E, k, dm, flags=_lib.RP_TOPK_DENSE)
    assert torch.equal(ids, ids2) and torch.equal(sc, sc2) and torch.equal(cnt, cnt2)


def test_sim_topk_exact_ties_and_order(gen):
    """Small-integer embeddings: every dot product is exact, many scores tie exactly, so ids must
    equal the oracle's (score desc, id asc) order bit for bit."""
    rng = np.random.default_rng(5)
    B, N, D, k = 40, 30000, 64, 100
    E = torch.from_numpy(rng.integers(-2, 3, size=(N, D)).astype(np.float32)).cuda().to(torch.bfloat16)
    Q = torch.from_numpy(rng.integers(-2, 3, size=(B, D)).astype(np.float32)).cuda().to(torch.bfloat16)
    m, acc = hh.synth_masks(rng, N, B, F=500)
    S = (Q.float() @ E.float().T).cpu().numpy()
    want_i, want_s = common_ref.masked_topk(S, acc, k)
    for flags in (_lib.RP_TOPK_AUTO, _lib.RP_TOPK_DENSE):
        ids, sc, cnt = hh.sim_topk(Q, E, k, hh.masks_to_device(m, Q.device), flags=flags)
        assert np.array_equal(ids.cpu().numpy(), want_i)
        assert np.array_equal(sc.cpu().numpy(), want_s)
        assert (cnt == k).all()


def test_sim_topk_random_shapes_exact(gen):
    """A seeded sweep over shapes nobody picked by hand - B, N, D, k, mask density, id offset - on small-integer embeddings
    (every product exact, ties everywhere): ids, scores and counts must equal the oracle's masked top-k bit for bit on every
    plan the library may choose, incl. rows with fewer than k accessible premises."""
    rng = np.random.default_rng(2027)
    for case in range(36):
        B = int(rng.choice([1, 2, 31, 64, 129, 200, 257, 600, 1024]))  # >= 512 queries: the small-list shape of the per-query stages
        N = int(rng.choice([1, 17, 255, 256, 257, 1000, 4097, 12345, 33000]))
        D = int(rng.choice([32, 64, 96, 128, 192, 1472]))
        k = int(rng.choice([1, 2, 10, 100, 333]))
        density = float(rng.choice([0.02, 0.3, 0.9]))
        off = int(rng.choice([0, 5000]))
        E = torch.from_numpy(rng.integers(-2, 3, size=(N, D)).astype(np.float32)).cuda().to(torch.bfloat16)
        Q = torch.from_numpy(rng.integers(-2, 3, size=(B, D)).astype(np.float32)).cuda().to(torch.bfloat16)
        m, acc = hh.synth_masks(rng, N, B, F=max(1, min(N, 40)), density=density)
        S = (Q.float() @ E.float().T).cpu().numpy()
        n_acc = np.minimum(acc.sum(1), k)
        want = [common_ref.masked_topk(S[b : b + 1], acc[b : b + 1], int(n_acc[b])) for b in range(B)]
        for flags in (_lib.RP_TOPK_AUTO, _lib.RP_TOPK_DENSE):
            ids, sc, cnt = hh.sim_topk(Q, E, k, hh.masks_to_device(m, Q.device), id_offset=off, flags=flags)
            ids, sc, cnt = ids.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy()
            assert np.array_equal(cnt, n_acc), (case, B, N, D, k)
            for b in range(B):
                c = int(n_acc[b])
                assert np.array_equal(ids[b, :c], want[b][0][0] + off), (case, B, N, D, k, b, flags)
                assert np.array_equal(sc[b, :c], want[b][1][0]), (case, B, N, D, k, b, flags)


def test_sim_topk_sample_gives_no_bound(gen):
    """The adversarial case for the two-pass plan: every SAMPLED block (rows [256 j stride, +256)) is
    inaccessible, so the sample yields no bound (thr = 0) and every accessible score of the other blocks is a
    candidate - tens of thousands per query instead of ~k*stride.  The two-pass plan keeps a BOUNDED candidate list
    (max(8192, 8 k stride) + k keys per query): such a query is reported as out_count = -1 and the caller repeats the
    search with RP_TOPK_DENSE (the ABI's contract, honoured by every product caller and by hh.sim_topk) - the answer
    must stay exact, for both filter generations, and for the sharded form."""
    rng = np.random.default_rng(31)
    B, N, D, k = 9, 70000, 128, 50
    E, Q = _rand_bf16(gen, N, D), _rand_bf16(gen, B, D)
    stride = 2
    while stride * 2 <= 64 and (stride * 2) ** 2 * k * 4 <= N:  # plan_sim's sampling stride (rp_retrieval.hip)
        stride *= 2
    rows = np.arange(N)
    sampled = (rows // 256) % stride == 0
    # one file per 256-row block; queries import every block except the sampled ones (and own none)
    file_of = (rows // 256).astype(np.int32)
    F = int(file_of.max()) + 1
    end_key = np.zeros(N, dtype=np.int64)
    own = np.full(B, F, dtype=np.int32)  # a file index no premise has
    qk = np.zeros(B, dtype=np.int64)
    imp = np.ones((B, F + 1), dtype=bool)
    imp[:, np.arange(0, F, stride)] = False
    imp[:, F] = False
    imp[3, :] = False          # one query with nothing accessible at all
    imp[4, 5:] = False         # one with a handful of blocks only
    words = (B + 31) // 32
    padded = np.zeros((F + 1, words * 32), dtype=np.uint8)
    padded[:, :B] = imp.T
    bits_t = np.packbits(padded, axis=1, bitorder="little").view(np.uint32).reshape(F + 1, words)
    acc = imp[:, file_of]
    assert not acc[:, sampled].any() and acc[0].sum() > 8192 + k
    S = _scores(Q, E)
    dm = hh.masks_to_device((file_of, end_key, bits_t, own, qk), Q.device)
    lib = _lib.load()
    for force_new in (1, 0):  # both filter generations (batches <= 128 default to the first)
        _lib.check(lib.rp_set_option(b"scan_force_new", force_new), "opt")
        try:
            raw = hh.sim_topk(Q, E, k, dm, retry_dense=False)[2].cpu().numpy()
            ids, sc, cnt = hh.sim_topk(Q, E, k, dm)
        finally:
            _lib.check(lib.rp_set_option(b"scan_force_new", 0), "opt")
        assert raw[0] == -1 and raw[3] == 0 and raw[4] >= 0, "the overflowing query is reported, the small ones are not"
        assert (cnt.cpu().numpy() >= 0).all()
        hh.check_topk_against_scores(ids.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy(), S, acc, k, tol=1e-4)
    ids2, sc2, cnt2 = hh.sim_topk(Q, E, k, dm, flags=_lib.RP_TOPK_DENSE)
    assert torch.equal(ids, ids2) and torch.equal(sc, sc2) and torch.equal(cnt, cnt2)
    # the sharded form of the same search (dist.hip_local_topk per rank + rp_topk_merge): a shard whose
    # sample gives no bound must still contribute its candidates to the merge
    f, ek, bt, own_d, qk_d = dm
    cut = 256 * stride * 8  # shard boundary on a sampled block, so shard 1 is adversarial too
    parts = [hh.sim_topk(Q, E[lo:hi].contiguous(), k, (f[lo:hi].contiguous(), ek[lo:hi].contiguous(), bt, own_d, qk_d),
                         id_offset=lo) for lo, hi in ((0, cut), (cut, N))]
    assert all((p[2].cpu().numpy() >= 0).all() for p in parts)
    mi, ms, mc = hh.topk_merge(torch.stack([p[1] for p in parts]), torch.stack([p[0] for p in parts]),
                               torch.stack([p[2] for p in parts]))
    assert torch.equal(mi, ids) and torch.equal(ms, sc) and torch.equal(mc, cnt)


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("B,N,D,k", [(1, 130000, 1472, 100), (7, 60000, 1536, 100), (32, 45000, 192, 10), (19, 30000, 128, 100),
                                     (33, 30000, 1472, 100), (64, 130000, 1472, 100), (50, 20000, 192, 10), (65, 30000, 1536, 100)])
def test_sim_topk_small_query_tiles_are_the_same_bits(gen, fp8, B, N, D, k):
    """At most 32 queries (a single proof state): the 32-query sample / filter tiles of round 6 (32 x 64 on two waves, 32 x 128 on
    four waves with a 4-deep ring; e4m3 rows of 1472 bytes end half a k-tile early) against the 128-query tiles every batch ran on
    before (option scan_small_tiles = 0): identical ids, scores and counts, and the properties of an exact masked
    top-k.  33 .. 64 queries take the 64-query filter tile (eight waves, three stages); B = 65 stays on the 128-query tiles either way."""
    rng = np.random.default_rng(B + N + D)
    E, Q = _rand_bf16(gen, N, D, scale=D ** -0.5), _rand_bf16(gen, B, D, scale=D ** -0.5)
    m, acc = hh.synth_masks(rng, N, B, F=max(2, N // 40))
    dm = hh.masks_to_device(m, Q.device)
    lib = _lib.load()
    if fp8:
        E8, es = hh.quantize_e4m3(E)
        Q8, qs = hh.quantize_e4m3(Q)
        run = lambda: hh.sim_topk_fp8(Q8, qs, E8, es, k, dm, id_offset=7)  # noqa: E731
    else:
        run = lambda: hh.sim_topk(Q, E, k, dm, id_offset=7)  # noqa: E731
    a = run()
    try:
        _lib.check(lib.rp_set_option(b"scan_small_tiles", 0), "opt")
        b = run()
    finally:
        _lib.check(lib.rp_set_option(b"scan_small_tiles", 1), "opt")
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    if not fp8:
        hh.check_topk_against_scores(a[0].cpu().numpy() - 7, a[1].cpu().numpy(), a[2].cpu().numpy(), _scores(Q, E), acc, k,
                                     tol=1e-4)


@pytest.mark.parametrize("B,N,D,k", [(256, 50000, 1472, 100), (300, 33000, 64, 10), (1, 40000, 192, 100),
                                     (37, 70000, 128, 1000), (600, 20000, 256, 3)])
def test_sim_topk_filter_generations_agree(gen, B, N, D, k):
    """scan_impl = 1 (first-generation filter kernel: queries as MFMA rows, per-score slow path) and the default
    (pipelined 256 x 256 tile, premises as MFMA rows, compacted survivors) must return identical tensors."""
    rng = np.random.default_rng(B + N)
    E, Q = _rand_bf16(gen, N, D, scale=D ** -0.5), _rand_bf16(gen, B, D, scale=D ** -0.5)
    m, acc = hh.synth_masks(rng, N, B, F=max(2, N // 40))
    dm = hh.masks_to_device(m, Q.device)
    lib = _lib.load()
    _lib.check(lib.rp_set_option(b"scan_force_new", 1), "opt")  # batches <= 128 default to the first generation
    try:
        a = hh.sim_topk(Q, E, k, dm, id_offset=1000)
        _lib.check(lib.rp_set_option(b"scan_impl", 1), "opt")
        b = hh.sim_topk(Q, E, k, dm, id_offset=1000)
    finally:
        _lib.check(lib.rp_set_option(b"scan_impl", 0), "opt")
        _lib.check(lib.rp_set_option(b"scan_force_new", 0), "opt")
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    hh.check_topk_against_scores(a[0].cpu().numpy() - 1000, a[1].cpu().numpy(), a[2].cpu().numpy(), _scores(Q, E), acc, k,
                                 tol=1e-4)


def test_sim_topk_fewer_than_k_accessible(gen):
    rng = np.random.default_rng(9)
    B, N, D, k = 8, 20000, 64, 100
    E, Q = _rand_bf16(gen, N, D), _rand_bf16(gen, B, D)
    m, acc = hh.synth_masks(rng, N, B, F=400, density=0.004)  # ~ a few dozen accessible each
    S = _scores(Q, E)
    ids, sc, cnt = hh.sim_topk(Q, E, k, hh.masks_to_device(m, Q.device))
    assert (acc.sum(1) < k).any()
    hh.check_topk_against_scores(ids.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy(), S, acc, k, tol=1e-4)


def test_shard_merge_equals_single_shot(gen):
    """Row-sharded corpus (north_star: 8 shards): per-shard masked top-k with id_offset, then
    rp_topk_merge, must equal the single-GPU answer exactly on ids and scores."""
    rng = np.random.default_rng(11)
    B, N, D, k, R = 64, 50000, 128, 100, 8
    E = torch.nn.functional.normalize(torch.randn(N, D, generator=gen, device="cuda"), dim=1).to(torch.bfloat16)
    Q = torch.nn.functional.normalize(torch.randn(B, D, generator=gen, device="cuda"), dim=1).to(torch.bfloat16)
    m, acc = hh.synth_masks(rng, N, B, F=900)
    f, ek, bt, own, qk = hh.masks_to_device(m, Q.device)
    ids, sc, cnt = hh.sim_topk(Q, E, k, (f, ek, bt, own, qk))
    bounds = np.linspace(0, N, R + 1).astype(int)
    parts = [hh.sim_topk(Q, E[lo:hi].contiguous(), k, (f[lo:hi].contiguous(), ek[lo:hi].contiguous(), bt, own, qk),
                         id_offset=int(lo)) for lo, hi in zip(bounds[:-1], bounds[1:])]
    g_i = torch.stack([p[0] for p in parts])
    g_s = torch.stack([p[1] for p in parts])
    g_c = torch.stack([p[2] for p in parts])
    mi, ms, mc = hh.topk_merge(g_s, g_i, g_c)
    assert torch.equal(mi, ids) and torch.equal(ms, sc) and torch.equal(mc, cnt)


@pytest.mark.parametrize("variant", [26, 17, 16, 15, 0, 9, 30])
@pytest.mark.parametrize("M", [256, 768])
def test_gemm_fused_rmsnorm_pieces_all_variants(gen, variant, M):
    """Residual epilogue on the two planes + per-64-feature sums of squares, and the row-scaled
    store / GEGLU epilogues, for every tile configuration (the auto small-M switch disabled)."""
    lib = _lib.load()
    N, K = 1472, 384
    np_ = (N + 63) // 64
    _lib.check(lib.rp_set_option(b"gemm_skinny", 0), "opt")
    _lib.check(lib.rp_set_option(b"gemm_variant_all", variant), "opt")
    try:
        A, W = _rand_bf16(gen, M, K), _rand_bf16(gen, N, K, scale=K ** -0.5)
        planes = hh.split_planes(torch.randn(M, N, generator=gen, device="cuda"))
        x0 = hh.merge_planes(planes)
        ssp = torch.full((np_, M), float("nan"), device="cuda")  # slot-major
        _lib.check(lib.rp_dbg_gemm_fused(A.data_ptr(), W.data_ptr(), planes.data_ptr(), M, N, K, N, _lib.RP_EPI_RESID, None,
                                         0, 0.0, 0.0, None, ssp.data_ptr(), np_, _lib.current_stream()), "fused")
        torch.cuda.synchronize()
        x, xb = hh.merge_planes(planes), planes[0]
        ref = x0 + A.float() @ W.float().T
        assert (x - ref).abs().max().item() < 2e-4
        # hi (the next GEMM's operand) is the bf16 rounding of x up to ties created by lo's own rounding
        assert ((x - xb.float()).abs() <= 2 ** -8 * xb.float().abs() + 1e-30).all()
        assert (xb != x.to(torch.bfloat16)).float().mean().item() < 1e-2
        want = (x.double() ** 2).view(M, np_, 64).sum(-1).float().T
        assert not torch.isnan(ssp).any(), "a sum-of-squares slot was never written"
        assert (ssp - want).abs().max().item() <= 1e-4 * want.abs().max().item()
        # consumer side: row-scaled store
        K2, N2 = N, 1152
        W2 = _rand_bf16(gen, N2, K2, scale=K2 ** -0.5)
        out = torch.empty(M, N2, dtype=torch.bfloat16, device="cuda")
        _lib.check(lib.rp_dbg_gemm_fused(xb.data_ptr(), W2.data_ptr(), out.data_ptr(), M, N2, K2, N2, _lib.RP_EPI_STORE_BF16,
                                         ssp.data_ptr(), np_, 1.0 / N, 1e-6, None, None, 0, _lib.current_stream()), "fused")
        torch.cuda.synchronize()
        rs = torch.rsqrt(ssp.sum(0) / N + 1e-6)
        ref2 = (xb.float() @ W2.float().T) * rs[:, None]
        assert (out.float() - ref2).abs().max().item() <= 2 ** -8 * ref2.abs().max().item() + 1e-3
    finally:
        _lib.check(lib.rp_set_option(b"gemm_skinny", 1), "opt")
        _lib.check(lib.rp_set_option(b"gemm_variant_all", -1), "opt")


def test_few_token_gemm_pipelined_loop_and_prefetch_helpers_change_no_bit(gen):
    """Passes of one proof state run the 64 x 128 x 64 tile on the software-pipelined loop (variant 17), and the launch's
    surplus workgroups prefetch the weight rows into the consumers' L2 (gemm_helpers).  Neither may change a bit of the
    result: same MFMA chain per element as the plain loop (variant 16), and helpers only move cache lines.  Token rows
    beyond the valid count are read as copies of the last valid row: the valid rows cannot notice."""
    lib = _lib.load()
    M, N, K = 256, 1152, 1472  # a QKV projection: 3.4 MB of weights, 36 workgroups -> helpers apply
    A, W = _rand_bf16(gen, M, K), _rand_bf16(gen, N, K, scale=K ** -0.5)
    ref = A.float() @ W.float().T
    res = []
    try:
        for opts in ({"gemm_small_pipe": 1, "gemm_helpers": 64}, {"gemm_small_pipe": 1, "gemm_helpers": 0},
                     {"gemm_small_pipe": 0, "gemm_helpers": 0}, {"gemm_small_pipe": 0, "gemm_helpers": 248}):
            for k_, v_ in opts.items():
                _lib.check(lib.rp_set_option(k_.encode(), v_), "opt")
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            _lib.check(lib.rp_dbg_gemm(A.data_ptr(), W.data_ptr(), out.data_ptr(), M, N, K, N, _lib.RP_EPI_STORE_BF16,
                                       _lib.current_stream()), "gemm")
            torch.cuda.synchronize()
            res.append(out)
    finally:
        _lib.check(lib.rp_set_option(b"gemm_small_pipe", 1), "opt")
        _lib.check(lib.rp_set_option(b"gemm_helpers", 64), "opt")
    assert (res[0].float() - ref).abs().max().item() <= 2 ** -8 * ref.abs().max().item() + 1e-3
    for other in res[1:]:
        assert torch.equal(res[0].view(torch.int16), other.view(torch.int16))


@pytest.mark.parametrize("B,N", [(2048, 16250), (1024, 32500), (512, 65000)])
def test_sim_topk_multi_gpu_shard_shape(gen, B, N):
    """The per-rank call of the 8- / 4- / 2-GPU bench: all world x 256 queries against one row shard of the 130 k index
    (>= 512 queries: the per-query stages take their small-list shape where the lists allow it); the dense plan must
    give the same bits."""
    rng = np.random.default_rng(21)
    D, k = 1472, 100
    E = torch.nn.functional.normalize(torch.randn(N, D, generator=gen, device="cuda"), dim=1).to(torch.bfloat16)
    Q = torch.nn.functional.normalize(torch.randn(B, D, generator=gen, device="cuda"), dim=1).to(torch.bfloat16)
    m, acc = hh.synth_masks(rng, N, B, F=600)
    ids, sc, cnt = hh.sim_topk(Q, E, k, hh.masks_to_device(m, Q.device), id_offset=5 * N)
    S = (Q.float() @ E.float().T).cpu().numpy()
    hh.check_topk_against_scores(ids.cpu().numpy() - 5 * N, sc.cpu().numpy(), cnt.cpu().numpy(), S, acc, k, tol=2e-5)
    ids2, sc2, cnt2 = hh.sim_topk(Q, E, k, hh.masks_to_device(m, Q.device), id_offset=5 * N, flags=_lib.RP_TOPK_DENSE)
    assert torch.equal(ids, ids2) and torch.equal(sc, sc2) and torch.equal(cnt, cnt2)


def test_build_file_bits_equals_host_transposition():
    """rp_build_file_bits (the per-batch accessibility operand built from the device-resident import closure and 4
    bytes per query) against Corpus.query_masks (the host-side transposition it replaces), incl. B not a multiple of 32."""
    import tempfile, os
    from reprover_amd import synth
    from reprover_amd.common import Context, Corpus, Pos

    files = synth.synth_corpus_records(70, 400, seed=9, max_imports=6)
    path = os.path.join(tempfile.mkdtemp(), "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    corpus = Corpus(path)
    rng = np.random.default_rng(4)
    for B in (1, 31, 32, 77):
        ctxs = [Context(files[int(f)]["path"], f"t{j}", Pos(int(rng.integers(1, 300)), 0), "a ⊢ b")
                for j, f in enumerate(rng.integers(0, len(files), size=B))]
        bits_t, own, qk = corpus.query_masks(ctxs)
        d_bits, d_own, d_qk = corpus.device_query_masks(ctxs, torch.device("cuda:0"))
        torch.cuda.synchronize()
        assert np.array_equal(d_bits.cpu().numpy().view(np.uint32), bits_t)
        assert np.array_equal(d_own.cpu().numpy(), own) and np.array_equal(d_qk.cpu().numpy(), qk)


def test_candidate_overflow_contract_through_the_product(gen):
    """Force the two-pass plan's candidate list to overflow for every query (option scan_cap = 1: capacity k + 1) and
    go through the product entry points: Corpus.get_nearest_premises, the hipGraph single-query path and the sharded
    search must each detect out_count = -1 and deliver, by the dense retry, exactly the answer of an unforced run."""
    import os, tempfile
    from reprover_amd import synth
    from reprover_amd.common import Context, Corpus, Pos
    from reprover_amd.dist import IndexShard, hip_local_topk
    from reprover_amd.retrieval.model import PremiseRetriever

    files = synth.synth_corpus_records(60, 40000, seed=77, max_imports=6)
    path = os.path.join(tempfile.mkdtemp(), "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    corpus = Corpus(path)
    N, D, B, k = len(corpus), 128, 5, 20
    E = torch.nn.functional.normalize(torch.randn(N, D, generator=gen, device="cuda"), dim=1).to(torch.bfloat16)
    Q = torch.nn.functional.normalize(torch.randn(B, D, generator=gen, device="cuda"), dim=1).to(torch.bfloat16)
    ctxs = [Context(files[50 + j]["path"], f"t{j}", Pos(400, 0), "a ⊢ b") for j in range(B)]
    lib = _lib.load()
    want = corpus.nearest_premise_ids(E, ctxs, Q, k)
    shard = IndexShard(corpus, np.array([0, N]), 0, torch.device("cuda:0"))
    shard.embeddings = E
    cfg = synth.t5_config("tiny")
    model = PremiseRetriever.from_state_dict(cfg, synth.synth_state_dict(cfg), 256, "cuda:0")
    model.corpus, model.embeddings_staled = corpus, False
    model.corpus_embeddings = torch.nn.functional.normalize(torch.randn(N, 128, generator=gen, device="cuda"), dim=1).to(torch.bfloat16)
    ref_single = model.retrieve("a ⊢ b", files[55]["path"], "t", Pos(400, 0), k)
    _lib.check(lib.rp_set_option(b"scan_cap", 1), "opt")
    try:
        raw = corpus.nearest_premise_ids(E, ctxs, Q, k)
        assert (raw[2] == -1).all(), "every query overflows its k + 1 slots"
        prem, scores = corpus.get_nearest_premises(E, ctxs, Q, k)  # PendingSearch.finish: dense retry
        assert [[p.full_name for p in row] for row in prem] == \
            [[corpus.all_premises[i].full_name for i in row] for row in want[0].cpu().tolist()]
        assert np.array_equal(np.array(scores, dtype=np.float32), want[1].cpu().numpy())
        ids, sc, cnt = hip_local_topk(shard, ctxs, Q, k)  # dist: the synchronous form retries densely
        assert torch.equal(ids, want[0]) and torch.equal(sc, want[1]) and torch.equal(cnt, want[2])
        model._drop_derived()
        got_single = model.retrieve("a ⊢ b", files[55]["path"], "t", Pos(400, 0), k)  # graph path -> fallback
        assert [p.full_name for p in got_single[0]] == [p.full_name for p in ref_single[0]]
        assert np.allclose(got_single[1], ref_single[1], atol=1e-6)
    finally:
        _lib.check(lib.rp_set_option(b"scan_cap", 0), "opt")


def test_mfma_probe_runs_and_validates_its_arguments():
    """rp_dbg_mfma_probe (the bench line's box calibration): launches, touches nothing but its sink, rejects bad shapes."""
    lib = _lib.load()
    sink = torch.zeros(4, device="cuda")
    _lib.check(lib.rp_dbg_mfma_probe(8, 64, sink.data_ptr(), _lib.current_stream()), "rp_dbg_mfma_probe")
    torch.cuda.synchronize()
    assert float(sink.abs().sum()) == 0.0
    assert lib.rp_dbg_mfma_probe(6, 64, sink.data_ptr(), _lib.current_stream()) != 0
    assert lib.rp_dbg_mfma_probe(8, 40, sink.data_ptr(), _lib.current_stream()) != 0
    assert lib.rp_dbg_mfma_probe(8, 64, None, _lib.current_stream()) != 0
