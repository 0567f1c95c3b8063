"""e4m3 index (BASELINE.json configs[4]): rp_quantize_rows_e4m3 and rp_sim_topk_fp8 through the C
ABI against oracle/fp8_ref.py.  Quantisation is integer work: bit-exact.  Scores: the MFMA sums exact
products in fp32, so |score - oracle| <= a few fp32 ulps of the accumulated magnitude (tolerance in
each test); with small-integer operands everything is exact and ids/scores must match bit for bit."""
import numpy as np
import pytest
import torch

import hip_helpers as hh
from oracle import common_ref, fp8_ref
from reprover_amd import _lib
from reprover_amd.common import Fp8Index

pytestmark = pytest.mark.gpu


def _unit_rows(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_quantize_rows_bit_exact(dtype):
    rng = np.random.default_rng(3)
    X = _unit_rows(rng, 300, 1536)
    X[7] = 0.0                      # all-zero row: scale 1, codes 0
    X[8] = 0.0
    X[8, 5] = -2.5                  # single entry: saturates to -448 exactly
    X[9, :64] *= 1e-4               # deep subnormals of the row's grid
    vals = fp8_ref.decode_e4m3(np.arange(0, 0x7F, dtype=np.uint8))
    mids = ((vals[:-1].astype(np.float64) + vals[1:].astype(np.float64)) / 2).astype(np.float32)
    X[10, :] = 0.0
    X[10, 0] = 448.0                # amax = 448 -> inv = 1: the remaining entries hit exact rounding ties
    X[10, 1:1 + len(mids)] = mids
    X[10, 200:200 + len(mids)] = -mids
    Xt = torch.from_numpy(X).cuda().to(dtype)
    codes, scale = hh.quantize_e4m3(Xt)
    want_c, want_s = fp8_ref.quantize_rows_e4m3(Xt.float().cpu().numpy())
    assert np.array_equal(scale.cpu().numpy(), want_s)
    assert np.array_equal(codes.cpu().numpy(), want_c)


def test_quantize_ragged_width_and_many_rows():
    rng = np.random.default_rng(4)
    for rows, D in [(1, 4), (5, 68), (1031, 1472)]:
        X = rng.standard_normal((rows, D)).astype(np.float32) * rng.uniform(1e-3, 10.0, (rows, 1)).astype(np.float32)
        codes, scale = hh.quantize_e4m3(torch.from_numpy(X).cuda())
        want_c, want_s = fp8_ref.quantize_rows_e4m3(X)
        assert np.array_equal(codes.cpu().numpy(), want_c) and np.array_equal(scale.cpu().numpy(), want_s)


@pytest.mark.parametrize("B", [40, 200])
def test_sim_topk_fp8_exact_small_integers(B):
    """Operands in {-2..2} (exact e4m3 codes), power-of-two scales: every score is exact, ties abound;
    ids and scores must equal the oracle's (score desc, id asc) order bit for bit, on both paths."""
    rng = np.random.default_rng(5)
    N, D, k = 30000, 128, 100
    Ei = rng.integers(-2, 3, size=(N, D)).astype(np.float32)
    Qi = rng.integers(-2, 3, size=(B, D)).astype(np.float32)
    E8, Q8 = fp8_ref.encode_e4m3(Ei), fp8_ref.encode_e4m3(Qi)
    es = (2.0 ** rng.integers(-3, 1, size=N)).astype(np.float32)
    qs = (2.0 ** rng.integers(-2, 2, size=B)).astype(np.float32)
    m, acc = hh.synth_masks(rng, N, B, F=500)
    S = fp8_ref.scores_fp8(Q8, qs, E8, es)
    assert np.array_equal(S, ((Qi @ Ei.T) * qs[:, None]) * es[None, :])
    want_i, want_s = common_ref.masked_topk(S, acc, k)
    dev = torch.device("cuda")
    args = [torch.from_numpy(a).to(dev) for a in (Q8, qs, E8, es)]
    for flags in (_lib.RP_TOPK_AUTO, _lib.RP_TOPK_DENSE):
        ids, sc, cnt = hh.sim_topk_fp8(*args, k, hh.masks_to_device(m, dev), flags=flags)
        assert np.array_equal(ids.cpu().numpy(), want_i)
        assert np.array_equal(sc.cpu().numpy(), want_s)
        assert (cnt == k).all()


@pytest.mark.parametrize("B,N,D,tol", [(256, 60000, 1536, 2e-6), (3, 20000, 1472, 2e-6), (130, 16000, 64, 2e-5),
                                       (256, 60000, 1472, 2e-6), (300, 40000, 192, 1e-5), (130, 16000, 128, 2e-5),
                                       (40, 30000, 320, 1e-5)])
def test_sim_topk_fp8_vs_oracle_scores(B, N, D, tol):
    """ByT5-base width (1536), ByT5-small width (1472 = 23 x 64: odd K-step count), minimum width (one
    K-step; unit vectors of 64 entries have a few large products that absorb the small ones in the fp32
    accumulator, hence the wider tolerance there).  B > 128 at D = 1472 / 192: e4m3 rows that end HALF a 128-byte k-tile
    early take the pipelined filter with its half last tile (GemmCfg::KTAIL); the dense plan (first-generation kernel, plain
    K-ascending loop) must give the same bits.  D = 128 / 320: the eight-wave sample tile of round 6 (128-byte k-tiles in a
    3-deep ring) with ONE whole k-tile, and with two and a half at B <= 128 (first-generation filter behind it)."""
    rng = np.random.default_rng(6)
    k = 100
    E, Q = _unit_rows(rng, N, D), _unit_rows(rng, B, D)
    E8, es = hh.quantize_e4m3(torch.from_numpy(E).cuda())
    Q8, qs = hh.quantize_e4m3(torch.from_numpy(Q).cuda())
    m, acc = hh.synth_masks(rng, N, B, F=700)
    S = fp8_ref.scores_fp8(Q8.cpu().numpy(), qs.cpu().numpy(), E8.cpu().numpy(), es.cpu().numpy())
    ids, sc, cnt = hh.sim_topk_fp8(Q8, qs, E8, es, k, hh.masks_to_device(m, Q8.device))
    # fp32 accumulation of exact products: a handful of ulps of the partial sums (|sum| < 1 here)
    got = sc.cpu().numpy()
    err = np.abs(np.take_along_axis(S, ids.cpu().numpy().astype(np.int64), axis=1) - got).max()
    print(f"fp8 scan B={B} N={N} D={D}: max|score - oracle| = {err:.3e} (scores up to {got.max():.3f})")
    hh.check_topk_against_scores(ids.cpu().numpy(), got, cnt.cpu().numpy(), S, acc, k, tol=tol)
    ids2, sc2, cnt2 = hh.sim_topk_fp8(Q8, qs, E8, es, k, hh.masks_to_device(m, Q8.device), flags=_lib.RP_TOPK_DENSE)
    assert torch.equal(ids, ids2) and torch.equal(sc, sc2) and torch.equal(cnt, cnt2)
    # how far e4m3 moves the scores from the unquantised ones: stated, and bounded
    S_full = Q @ E.T
    assert np.abs(S - S_full).max() < (1e-2 if D >= 1024 else 5e-2)  # ~ 0.05 / sqrt(D) per quantised side


def test_fp8_shard_merge_equals_single_shot():
    rng = np.random.default_rng(7)
    B, N, D, k, R = 64, 48000, 256, 50, 8
    E, Q = _unit_rows(rng, N, D), _unit_rows(rng, B, D)
    E8, es = hh.quantize_e4m3(torch.from_numpy(E).cuda())
    Q8, qs = hh.quantize_e4m3(torch.from_numpy(Q).cuda())
    m, acc = hh.synth_masks(rng, N, B, F=300)
    f, ek, bt, own, qk = hh.masks_to_device(m, Q8.device)
    ids, sc, cnt = hh.sim_topk_fp8(Q8, qs, E8, es, k, (f, ek, bt, own, qk))
    bounds = np.linspace(0, N, R + 1).astype(int)
    parts = [hh.sim_topk_fp8(Q8, qs, E8[lo:hi].contiguous(), es[lo:hi].contiguous(), k,
                             (f[lo:hi].contiguous(), ek[lo:hi].contiguous(), bt, own, qk), id_offset=int(lo))
             for lo, hi in zip(bounds[:-1], bounds[1:])]
    mi, ms, mc = hh.topk_merge(torch.stack([p[1] for p in parts]), torch.stack([p[0] for p in parts]),
                               torch.stack([p[2] for p in parts]))
    assert torch.equal(mi, ids) and torch.equal(ms, sc) and torch.equal(mc, cnt)


def test_fp8_argument_errors():
    lib = _lib.load()
    dev = torch.device("cuda")
    Q8 = torch.zeros(2, 96, dtype=torch.uint8, device=dev)
    s = torch.ones(2, device=dev)
    with pytest.raises(_lib.HipLibraryError, match="multiple of 64"):
        hh.sim_topk_fp8(Q8, s, Q8, s, 1)
    Q8 = torch.zeros(2, 64, dtype=torch.uint8, device=dev)
    st = lib.rp_sim_topk_fp8(_lib.ptr(Q8), None, _lib.ptr(Q8), None, 2, 2, 64, None, None, None, 0, None, None, 0, 1,
                             0, _lib.ptr(s), _lib.ptr(s), _lib.ptr(s), _lib.ptr(s), 8, None)
    assert st == -1 and b"scale" in lib.rp_last_error()
    with pytest.raises(_lib.HipLibraryError, match="multiple of 4"):
        hh.quantize_e4m3(torch.zeros(2, 6, device=dev))


def test_host_fp8_index_search_and_value_error():
    """Corpus.get_nearest_premises with an Fp8Index: same premises/scores as the oracle's masked top-k on
    the quantised operands; ValueError when a query has fewer than k accessible premises."""
    import json, os, tempfile
    from reprover_amd.common import Context, Corpus, Pos
    from reprover_amd import synth

    rng = np.random.default_rng(8)
    recs = synth.synth_corpus_records(n_files=60, n_premises=3000, seed=11)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "corpus.jsonl")
        with open(path, "w") as fh:
            for r in recs:
                fh.write(json.dumps(r) + "\n")
        corpus = Corpus(path)
    N, D, k = len(corpus.all_premises), 192, 10
    E, Q = _unit_rows(rng, N, D), _unit_rows(rng, 6, D)
    idx = Fp8Index.quantize(torch.from_numpy(E).cuda())
    files = [f.path for f in corpus.files]
    ctxs = [Context(files[-1 - i], "T", Pos(10 ** 6, 0), "⊢ s") for i in range(6)]
    prem, scores = corpus.get_nearest_premises(idx, ctxs, torch.from_numpy(Q).cuda(), k)
    Q8, qs = fp8_ref.quantize_rows_e4m3(Q)
    S = fp8_ref.scores_fp8(Q8, qs, idx.codes.cpu().numpy(), idx.scale.cpu().numpy())
    acc = np.stack([corpus.accessible_mask(c.path, c.theorem_pos) for c in ctxs])
    want_i, want_s = common_ref.masked_topk(S, acc, k)
    got_i = np.array([[corpus.all_premises.index(p) for p in row] for row in prem])
    assert np.abs(np.array(scores) - want_s).max() < 1e-5  # D = 192: fp32 accumulation of a few large products
    # ids must agree wherever the oracle's score is separated from both neighbours by more than 2 x the score
    # tolerance (the rule of every other id comparison; swaps are only possible between closer scores)
    checked, bad = hh.gap_rule_ids(got_i.tolist(), want_i.tolist(), want_s.tolist(), tol=1e-5)
    assert checked > 0.8 * want_i.size and bad == 0, (checked, bad)
    assert np.allclose(idx.dequantize().cpu().numpy(), fp8_ref.decode_e4m3(idx.codes.cpu().numpy()) * idx.scale.cpu().numpy()[:, None])
    first = Context(files[0], "T", Pos(0, 0), "⊢ s")  # nothing before it, nothing imported
    with pytest.raises(ValueError):
        corpus.get_nearest_premises(idx, [first], torch.from_numpy(Q[:1]).cuda(), k)


def test_retriever_fp8_index_c5_slice():
    """BASELINE configs[4] in miniature: ByT5-base geometry (d_model 1536, 12 heads, d_ff 3968; 3 layers)
    + e4m3 index.  reindex -> quantised copy -> retrieve / predict_step; the fp8 results must be the
    oracle's masked top-k on the quantised operands, and agree with the bf16 index on most of the top-10."""
    import os, tempfile
    from reprover_amd import synth
    from reprover_amd.common import Context, Pos
    from reprover_amd.retrieval.model import PremiseRetriever

    cfg = synth.t5_config("byt5-base")
    cfg["num_layers"] = 3
    sd = synth.synth_state_dict(cfg, seed=5)
    model = PremiseRetriever.from_state_dict(cfg, sd, 512, "cuda:0")
    files = synth.synth_corpus_records(40, 1500, seed=13)
    path = os.path.join(tempfile.mkdtemp(), "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    model.load_corpus(path)
    model.reindex_corpus(64)
    corpus = model.corpus
    rng = np.random.default_rng(2)
    last = corpus.files[-1].path
    ctxs = [Context(last, f"t{j}", Pos(10 ** 6, 0), "h : " + synth.synth_text(rng, 60 + 17 * j) + " ⊢ goal") for j in range(8)]
    k = 10
    where = {id(p): i for i, p in enumerate(corpus.all_premises)}
    bf = [model.retrieve(c.state, c.path, c.theorem_full_name, c.theorem_pos, k) for c in ctxs]
    model.index_dtype = "fp8"
    f8 = [model.retrieve(c.state, c.path, c.theorem_full_name, c.theorem_pos, k) for c in ctxs]
    idx = model._search_operand()
    assert isinstance(idx, Fp8Index) and idx.shape == (len(corpus.all_premises), 1536)
    assert model.corpus_embeddings.dtype == torch.bfloat16  # what the reference exposes is untouched
    # oracle on the quantised operands
    Qe = torch.cat([model.encode_texts([c.serialize()]) for c in ctxs])  # one state per pass, as retrieve() encodes
    Q8, qs = fp8_ref.quantize_rows_e4m3(Qe.float().cpu().numpy())
    S = fp8_ref.scores_fp8(Q8, qs, idx.codes.cpu().numpy(), idx.scale.cpu().numpy())
    acc = np.stack([corpus.accessible_mask(c.path, c.theorem_pos) for c in ctxs])
    want_i, want_s = common_ref.masked_topk(S, acc, k)
    got_i = np.array([[where[id(p)] for p in prem] for prem, _ in f8])
    got_s = np.array([sc for _, sc in f8])
    assert np.abs(got_s - want_s).max() < 5e-6
    checked, bad = hh.gap_rule_ids(got_i.tolist(), want_i.tolist(), want_s.tolist(), tol=5e-6)
    assert bad == 0
    overlap = np.mean([len({where[id(p)] for p in a[0]} & {where[id(p)] for p in b[0]}) / k for a, b in zip(bf, f8)])
    dscore = max(abs(x - y) for a, b in zip(bf, f8) for x, y in zip(sorted(a[1]), sorted(b[1])))
    print(f"c5 slice: fp8 vs bf16 index: top-{k} overlap {overlap:.3f}, max |sorted score diff| {dscore:.3e}")
    assert overlap >= 0.8 and dscore < 2e-2
    # batch path (predict_step) uses the same operand
    model.num_retrieved = k
    tok = model.tokenizer([c.serialize() for c in ctxs], padding="longest", max_length=512, truncation=True,
                          return_tensors="pt")
    b = {"context": ctxs, "context_ids": tok.input_ids.cuda(), "context_mask": tok.attention_mask.cuda()}
    for key in ("url", "commit", "file_path", "full_name", "start", "tactic_idx", "all_pos_premises"):
        b[key] = [None] * len(ctxs)
    model.predict_step_outputs = []
    model.predict_step(b, 0)
    # (ids agree under the gap rule)
    batch_i = [[where[id(p)] for p in r["retrieved_premises"]] for r in model.predict_step_outputs]
    checked, bad = hh.gap_rule_ids(batch_i, want_i.tolist(), want_s.tolist(), tol=4e-3)
    assert bad == 0 and checked > 0


def test_fp8_full_size_properties_1m_d1536():
    """BASELINE configs[4] at its full index size: 1,000,000 premises x 1536 (ByT5-base width) in e4m3,
    B = 256, k = 100.  The oracle does not finish this size in seconds, so: scores/top-k against a plain
    torch fp32 evaluation of the same quantised operands, and the size-independent properties — sorted,
    only accessible premises, count = k, dense pass == two-pass, 8-way shard + merge == single shot."""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(2024)
    rng = np.random.default_rng(2024)
    N, D, B, k, F = 1_000_000, 1536, 256, 100, 5000
    E8 = torch.empty((N, D), dtype=torch.uint8, device="cuda")
    es = torch.empty((N,), dtype=torch.float32, device="cuda")
    for lo in range(0, N, 125_000):
        x = torch.nn.functional.normalize(torch.randn(125_000, D, generator=gen, device="cuda"), dim=1)
        c, s_ = hh.quantize_e4m3(x)
        E8[lo : lo + 125_000], es[lo : lo + 125_000] = c, s_
    Q8, qs = hh.quantize_e4m3(torch.nn.functional.normalize(torch.randn(B, D, generator=gen, device="cuda"), dim=1))
    m, acc = hh.synth_masks(rng, N, B, F)
    dm = hh.masks_to_device(m, Q8.device)
    ids, sc, cnt = hh.sim_topk_fp8(Q8, qs, E8, es, k, dm)
    S = np.empty((B, N), dtype=np.float32)
    qf = Q8.view(torch.float8_e4m3fn).float()
    for lo in range(0, N, 125_000):
        part = (qf @ E8[lo : lo + 125_000].view(torch.float8_e4m3fn).float().T) * qs[:, None] * es[None, lo : lo + 125_000]
        S[:, lo : lo + 125_000] = part.cpu().numpy()
    hh.check_topk_against_scores(ids.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy(), S, acc, k, tol=5e-6)
    assert (cnt == k).all()
    ids2, sc2, cnt2 = hh.sim_topk_fp8(Q8, qs, E8, es, k, dm, flags=_lib.RP_TOPK_DENSE)
    assert torch.equal(ids, ids2) and torch.equal(sc, sc2) and torch.equal(cnt, cnt2)
    f, ek, bt, own, qk = dm
    bounds = np.linspace(0, N, 9).astype(int)
    parts = [hh.sim_topk_fp8(Q8, qs, E8[lo:hi], es[lo:hi], k, (f[lo:hi], ek[lo:hi], bt, own, qk), id_offset=int(lo))
             for lo, hi in zip(bounds[:-1], bounds[1:])]
    mi, ms, mc = hh.topk_merge(torch.stack([p[1] for p in parts]), torch.stack([p[0] for p in parts]),
                               torch.stack([p[2] for p in parts]))
    assert torch.equal(mi, ids) and torch.equal(ms, sc) and torch.equal(mc, cnt)
