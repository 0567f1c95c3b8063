"""End-to-end parity of the drop-in classes on the GPU: Corpus.get_nearest_premises (G6),
reindex_corpus + predict_step + retrieve on BASELINE config 1 (G7: 1k premises, 128 states,
top-10, ByT5-small), and size-independent properties at BASELINE config 2 size (130k premises,
B=256, k=100)."""
import json
import os
import pickle
import tempfile

import numpy as np
import pytest
import torch

import hip_helpers as hh
from oracle import common_ref
from oracle import parity_margins as pm
from reprover_amd import _lib, synth
from reprover_amd.common import Context, Corpus, IndexedCorpus, Pos
from reprover_amd.retrieval.model import PremiseRetriever

pytestmark = pytest.mark.gpu


def _bf16_round(a: np.ndarray) -> np.ndarray:
    return torch.from_numpy(a).to(torch.bfloat16).float().numpy()


def test_g6_get_nearest_premises(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "g6_nearest.json")))
    z = np.load(os.path.join(golden_dir, "g6_nearest.npz"))
    files = synth.synth_corpus_records(g["n_files"], g["n_premises"], seed=g["corpus_seed"])
    path = os.path.join(tempfile.mkdtemp(), "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    corpus, ref = Corpus(path), common_ref.CorpusRef(path)
    E, Q = torch.from_numpy(z["E"]).cuda(), torch.from_numpy(z["Q"]).cuda()
    Eb, Qb = _bf16_round(z["E"]), _bf16_round(z["Q"])  # what the kernel sees
    ctxs = [Context(q["path"], f"thm{j}", Pos(*q["pos"]), f"x{j} ⊢ y") for j, q in enumerate(g["queries"])]
    rctx = [common_ref.ContextRef(q["path"], f"thm{j}", common_ref.Pos(*q["pos"]), f"x{j} ⊢ y")
            for j, q in enumerate(g["queries"])]
    where = {id(p): i for i, p in enumerate(corpus.all_premises)}
    for k, res in g["results"].items():
        qs = res["queries"]
        prem, scores = corpus.get_nearest_premises(E, [ctxs[j] for j in qs], Q[qs], int(k))
        ids = [[where[id(p)] for p in row] for row in prem]
        assert all(isinstance(s, float) for row in scores for s in row)
        # (a) exact against the oracle evaluated on the same bf16-rounded operands
        oid, osc = ref.get_nearest_premises(Eb, [rctx[j] for j in qs], Qb[qs], int(k))
        checked, bad = hh.gap_rule_ids(ids, oid, osc, tol=1e-6)
        assert bad == 0 and (checked > 0 or int(k) == 1)
        assert np.abs(np.array(scores) - np.array(osc)).max() < 1e-5
        # (b) against the reference's fp32 golden output, within the stated bf16 tolerance
        checked, bad = hh.gap_rule_ids(ids, res["ids"], res["scores"], tol=1e-2)
        assert bad == 0
        assert np.abs(np.array(scores) - np.array(res["scores"])).max() < 1e-2
    with pytest.raises(ValueError):  # common.py:323-324
        bad_q = g["value_error_query"]
        corpus.get_nearest_premises(E, [ctxs[bad_q]], Q[[bad_q]], 100)


@pytest.fixture(scope="module")
def g7(golden_dir, small_weights):
    g = json.load(open(os.path.join(golden_dir, "g7_predict.json")))
    z = np.load(os.path.join(golden_dir, "g7_predict.npz"))
    files = synth.synth_corpus_records(g["n_files"], g["n_premises"], seed=g["corpus_seed"],
                                       code_bytes=tuple(g["code_bytes"]))
    path = os.path.join(tempfile.mkdtemp(), "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    cfg, sd = small_weights
    model = PremiseRetriever.from_state_dict(cfg, sd, g["max_seq_len"], "cuda:0")  # bf16, the GPU default
    model.load_corpus(path)
    assert model.embeddings_staled
    model.reindex_corpus(batch_size=g["batch_size"])
    assert not model.embeddings_staled
    return g, z, model, path


@pytest.fixture(scope="module")
def g7h(golden_dir, small_weights_hf):
    """Fixture G7h: BASELINE config 1 on HF-init-scale weights and the family corpus (near-duplicate premises, states
    built from a family's base text), whose fp32 top-10 scores are spread widely enough for the id gap rule to bite."""
    g = json.load(open(os.path.join(golden_dir, "g7h_predict.json")))
    z = np.load(os.path.join(golden_dir, "g7h_predict.npz"))
    path = os.path.join(tempfile.mkdtemp(), "corpus.jsonl")
    synth.write_corpus_jsonl(path, pm.g7_corpus_records(g))
    cfg, sd = small_weights_hf
    model = PremiseRetriever.from_state_dict(cfg, sd, g["max_seq_len"], "cuda:0")  # bf16, the GPU default
    model.load_corpus(path)
    model.reindex_corpus(batch_size=g["batch_size"])
    return g, z, model, path


def test_g7h_written_contract_reindex_predict_retrieve(g7h, parity_margins):
    """The written contract, un-relaxed, end to end on configs[0]: every re-indexed row's cosine with the reference's
    fp32 row >= 0.999; predict_step's scores within 1e-2 of the reference's; ids equal at EVERY rank whose fp32 score is
    more than 2 x tol away from both neighbours - and on this fixture that is at least a quarter of all ranks; and the
    engine at least as close to fp32 as the reference's own bf16 mode on the same inputs."""
    g, z, model, _ = g7h
    k = g["k"]
    assert len(model.corpus) == g["N"]
    mr = parity_margins["g7h_reindex_rows_bf16_hf_init"] = pm.g7_row_margins(model, g, z)
    print(f"g7h rows: {mr}")
    pm.assert_written_contract(mr)
    recs, ids, scores, ctxs, m = pm.g7_predict(model, g)
    parity_margins["g7h_predict_128_states_top10_hf_init"] = m
    print(f"g7h predict: {m}")
    pm.assert_written_contract(m)
    assert m["gap_rule_ranks_checked"] >= 320 and m["gap_rule_ranks_checked"] == g["gap_rule_ranks_checkable"]
    # ids beside the reference's own GPU mode (HF-bf16 embeddings + bf16 similarity matrix), both against fp32 and both
    # reported in the margins; where the fp32 scores are closer than the tolerance either may swap neighbours, so these two
    # statistics are held to HF-bf16's within one state / 1 % (the gap rule above is the exact id requirement)
    assert m["top1_agreement"] >= m["hf_bf16_top1_agreement"] - 1.0 / len(ctxs) - 1e-9
    assert m[f"top{k}_overlap"] >= m[f"hf_bf16_top{k}_overlap"] - 0.01
    where = {id(p): i for i, p in enumerate(model.corpus.all_premises)}
    for j, single in enumerate(g["retrieve"]):  # the single-query path (model.py:338-375)
        c = ctxs[j]
        prem, sc = model.retrieve(c.state, c.path, c.theorem_full_name, c.theorem_pos, k)
        got = [where[id(p)] for p in prem]
        checked1, bad1 = hh.gap_rule_ids([got], [single["ids"]], [single["scores"]], tol=1e-2)
        assert bad1 == 0
        assert np.abs(np.array(sc) - np.array(single["scores"])).max() <= 1e-2


def test_g7_reindex_corpus(g7, parity_margins):
    g, z, model, _ = g7
    E = model.corpus_embeddings
    assert E.shape == (g["N"], 1472) and E.dtype == torch.bfloat16 and E.is_cuda
    Ef = E.float().cpu()
    assert torch.allclose(Ef.norm(dim=1), torch.ones(g["N"]), atol=1e-2)
    cos = torch.nn.functional.cosine_similarity(Ef[:16], torch.from_numpy(z["E_head"]), dim=1)
    # HuggingFace's own bf16 mode (the reference's GPU numerics) reaches only this cosine with its
    # fp32 self on these sharp synthetic weights; the engine must be at least that close.
    assert cos.min().item() >= max(0.997, g["hf_bf16_min_embedding_cosine"])
    # every row against the reference's fp32 matrix (stored as fp16: 5e-4 relative, far below the bar)
    m = parity_margins["g7_reindex_1005_rows_bf16"] = pm.g7_row_margins(model, g, z)
    print(f"g7 rows: {m}")
    assert m["min_row_cosine"] >= g["hf_bf16_min_embedding_cosine"]  # no row further from fp32 than HF-bf16's worst
    assert m["mean_row_cosine"] >= 0.998  # bf16 rows (the GPU default dtype): measured 0.9985
    assert m["max_abs_emb_err"] <= 2e-2


def test_g7_predict_and_retrieve(g7, parity_margins):
    g, z, model, _ = g7
    k = g["k"]
    where = {id(p): i for i, p in enumerate(model.corpus.all_premises)}
    recs, ids, scores, ctxs, m = pm.g7_predict(model, g)
    parity_margins["g7_predict_128_states_top10"] = m
    print(f"g7 predict: {m}")
    assert len(recs) == len(ctxs) and set(recs[0]) == {
        "url", "commit", "file_path", "full_name", "start", "tactic_idx", "context", "all_pos_premises",
        "retrieved_premises", "scores"}
    # stated tolerance: 1e-2 abs, or what HuggingFace-bf16 itself needs on these inputs
    # if that is looser (fixture: the reference re-run in bf16, scores at the golden ids) - the relaxation
    # oracle/parity_margins.py labels; m["contract_met"] records whether 1e-2 holds as written
    hf_err = m["hf_bf16_max_abs_score_err"]
    assert m["max_abs_score_err"] <= max(1e-2, hf_err)
    assert m["gap_rule_mismatches"] == 0
    assert m[f"top{k}_overlap"] >= 0.9
    # single-query path (model.py:338-375)
    for j, single in enumerate(g["retrieve"]):
        c = ctxs[j]
        prem, sc = model.retrieve(c.state, c.path, c.theorem_full_name, c.theorem_pos, k)
        # one state alone: the same answer as inside the batch up to swaps between scores closer than the tolerance
        # (the gap rule of every id comparison), scores within the stated bound of the golden ones
        got = [where[id(p)] for p in prem]
        checked1, bad1 = hh.gap_rule_ids([got], [single["ids"]], [single["scores"]], tol=1e-2)
        assert bad1 == 0 and len(set(got) & set(ids[j])) >= k - 2
        assert np.abs(np.array(sc) - np.array(single["scores"])).max() <= max(1e-2, hf_err)
    # predictions.pickle round trip (model.py:329-336)
    d = tempfile.mkdtemp()
    model.on_predict_epoch_end(d)
    back = pickle.load(open(os.path.join(d, "predictions.pickle"), "rb"))
    assert len(back) == len(ctxs) and back[3]["retrieved_premises"][0].full_name == \
        model.corpus.all_premises[ids[3][0]].full_name


def test_predict_step_pipeline_order_and_late_errors(g7):
    """predict_step completes batch i-1 after launching batch i: records must come out in call order, identical to a
    run that reads the outputs after every call (no overlap), and the reference's ValueError (common.py:323-324)
    must still surface - at the latest when the outputs are read."""
    g, z, model, _ = g7
    k = g["k"]
    model.num_retrieved = k
    ctxs = [Context(q["path"], f"thm{j}", Pos(*q["pos"]), q["state"]) for j, q in enumerate(g["queries"][:24])]

    def make_batch(batch):
        tok = model.tokenizer([c.serialize() for c in batch], padding="longest", max_length=g["max_seq_len"],
                              truncation=True, return_tensors="pt")
        b = {"context": batch, "context_ids": tok.input_ids, "context_mask": tok.attention_mask}
        for key in ("url", "commit", "file_path", "full_name", "start", "tactic_idx", "all_pos_premises"):
            b[key] = [None] * len(batch)
        b["tactic_idx"] = list(range(len(batch)))
        return b

    batches = [make_batch(ctxs[i : i + 8]) for i in range(0, 24, 8)]
    model.predict_step_outputs = []
    for b in batches:  # pipelined: nothing is read between the calls
        model.predict_step(b, 0)
    piped = model.predict_step_outputs
    model.predict_step_outputs = []
    stepwise = []
    for b in batches:  # every call followed by a read: batch i is complete before batch i+1 is launched
        model.predict_step(b, 0)
        stepwise = list(model.predict_step_outputs)
    assert len(piped) == len(stepwise) == 24
    for a, b in zip(piped, stepwise):
        assert a["context"] is b["context"] and a["scores"] == b["scores"]
        assert [p.full_name for p in a["retrieved_premises"]] == [p.full_name for p in b["retrieved_premises"]]
    # fewer than k accessible premises: nothing accessible from the first file's first line
    first = model.corpus.files[0].path
    bad = make_batch([Context(first, "t", Pos(0, 0), "x ⊢ y")])
    model.predict_step_outputs = []
    model.predict_step(batches[0], 0)
    model.predict_step(bad, 0)  # launches; completes batches[0]
    with pytest.raises(ValueError):
        model.predict_step_outputs  # completing the bad batch raises
    model.predict_step_outputs = []


def test_predict_step_strict_mode_raises_in_the_submitting_call(g7):
    """``predict_pipeline = False``: the reference's synchronous semantics (retrieval/model.py:281-290 +
    common.py:323-324) - predict_step returns with its own batch's records appended, and a batch with fewer than k
    accessible premises raises ValueError in the very call that submitted it, leaving nothing queued or in flight."""
    g, z, model, _ = g7
    k = g["k"]
    model.num_retrieved = k
    ctxs = [Context(q["path"], f"thm{j}", Pos(*q["pos"]), q["state"]) for j, q in enumerate(g["queries"][:16])]

    def make_batch(batch):
        tok = model.tokenizer([c.serialize() for c in batch], padding="longest", max_length=g["max_seq_len"],
                              truncation=True, return_tensors="pt")
        b = {"context": batch, "context_ids": tok.input_ids, "context_mask": tok.attention_mask}
        for key in ("url", "commit", "file_path", "full_name", "start", "tactic_idx", "all_pos_premises"):
            b[key] = [None] * len(batch)
        return b

    batches = [make_batch(ctxs[i : i + 8]) for i in range(0, 16, 8)]
    model.predict_step_outputs = []
    for b in batches:
        model.predict_step(b, 0)
    piped = model.predict_step_outputs
    assert model.predict_pipeline is True
    model.predict_pipeline = False
    try:
        model.predict_step_outputs = []
        for n, b in enumerate(batches):
            model.predict_step(b, 0)
            # complete on return: the private list (no lazy completion through the property) already holds the records
            assert len(model._predict_outputs) == 8 * (n + 1) and model._predict_pending is None and not model._predict_stash
        strict = model.predict_step_outputs
        assert len(strict) == len(piped) == 16
        for a, b in zip(strict, piped):
            assert a["context"] is b["context"] and a["scores"] == b["scores"]
            assert [p.full_name for p in a["retrieved_premises"]] == [p.full_name for p in b["retrieved_premises"]]
        first = model.corpus.files[0].path
        bad = make_batch([Context(first, "t", Pos(0, 0), "x ⊢ y")])  # nothing accessible from the first file's first line
        with pytest.raises(ValueError):
            model.predict_step(bad, 0)  # raised HERE, not one call later
        assert model._predict_pending is None and len(model._predict_outputs) == 16
        model.predict_step(batches[0], 0)  # and the object is usable afterwards
        assert len(model._predict_outputs) == 24
    finally:
        model.predict_pipeline = True
        model.predict_step_outputs = []


def test_predict_step_coalesces_host_batches(g7):
    """Consecutive host batches run as one GPU pass once ``predict_coalesce_states`` states are queued: the same records
    in the same order as one pass per batch (a row's result does not depend on its pass), a device-tensor batch in
    between flushes what is queued first, reading the outputs completes everything."""
    g, z, model, _ = g7
    k = g["k"]
    model.num_retrieved = k
    ctxs = [Context(q["path"], f"thm{j}", Pos(*q["pos"]), q["state"]) for j, q in enumerate(g["queries"][:40])]

    def make_batch(batch, device=None):
        tok = model.tokenizer([c.serialize() for c in batch], padding="longest", max_length=g["max_seq_len"],
                              truncation=True, return_tensors="pt")
        b = {"context": batch, "context_ids": tok.input_ids, "context_mask": tok.attention_mask}
        if device is not None:
            b["context_ids"], b["context_mask"] = b["context_ids"].to(device), b["context_mask"].to(device)
        for key in ("url", "commit", "file_path", "full_name", "start", "tactic_idx", "all_pos_premises"):
            b[key] = [None] * len(batch)
        return b

    sizes = [8, 8, 5, 8, 8, 3]  # 40 states in ragged batches
    cuts = np.cumsum([0] + sizes)
    old = model.predict_coalesce_states
    try:
        runs = {}
        for coalesce in (0, 16, 256):
            model.predict_coalesce_states = coalesce
            model.predict_step_outputs = []
            for i in range(len(sizes)):
                model.predict_step(make_batch(ctxs[cuts[i] : cuts[i + 1]], "cuda" if (coalesce == 16 and i == 2) else None), 0)
                if coalesce == 16 and i == 1:  # 16 states queued: the pass is launched, nothing is complete yet
                    assert model._predict_stash == [] and model._predict_pending is not None and not model._predict_outputs
            if coalesce == 256:
                assert len(model._predict_stash) == len(sizes) and not model._predict_outputs  # all still queued
            runs[coalesce] = model.predict_step_outputs
            assert len(runs[coalesce]) == 40 and model._predict_stash == [] and model._predict_pending is None
        for coalesce in (16, 256):
            for a, b in zip(runs[0], runs[coalesce]):
                assert a["context"] is b["context"] and a["scores"] == b["scores"]
                assert [p.full_name for p in a["retrieved_premises"]] == [p.full_name for p in b["retrieved_premises"]]
    finally:
        model.predict_coalesce_states = old
        model.predict_step_outputs = []


def test_retrieve_graph_replay_equals_launch_by_launch(g7):
    """retrieve() as one hipGraph replay (single_query.py: padded encode + masked top-k on static buffers) must
    return exactly what the launch-by-launch path returns - same premises, same scores - for states of every
    length bucket, repeatedly (buffers are reused), and raise the reference's ValueError alike."""
    g, z, model, _ = g7
    rng = np.random.default_rng(5)
    ctxs = [(q["path"], f"thm{j}", Pos(*q["pos"])) for j, q in enumerate(g["queries"][:6])]
    states = [synth.synth_state(rng, n) for n in (9, 100, 127, 128, 300, 700, 1023, 1500)]
    assert model.use_graphs
    for rep in range(2):
        for j, st in enumerate(states):
            path, name, pos = ctxs[j % len(ctxs)]
            model.use_graphs = True
            a = model.retrieve(st, path, name, pos, 10)
            model.use_graphs = False
            b = model.retrieve(st, path, name, pos, 10)
            model.use_graphs = True
            assert [p.full_name for p in a[0]] == [p.full_name for p in b[0]] and a[1] == b[1], (rep, j)
    assert model._single_query is not None and len(model._single_query._graphs) >= 3  # several buckets captured
    first = model.corpus.files[0].path
    with pytest.raises(ValueError):  # nothing is accessible from the first file's first line
        model.retrieve(states[1], first, "t", Pos(0, 0), 10)
    # a new embedding matrix invalidates the captured graphs (they hold raw pointers into the old one)
    old = model.corpus_embeddings
    model._drop_derived()
    model.corpus_embeddings = old.flip(0).contiguous()
    path, name, pos = ctxs[0]
    a = model.retrieve(states[1], path, name, pos, 10)
    model.use_graphs = False
    b = model.retrieve(states[1], path, name, pos, 10)
    model.use_graphs = True
    assert [p.full_name for p in a[0]] == [p.full_name for p in b[0]] and a[1] == b[1]
    model._drop_derived()
    model.corpus_embeddings = old


def test_indexed_corpus_pickle_roundtrip(g7):
    g, z, model, _ = g7
    path = os.path.join(tempfile.mkdtemp(), "indexed.pickle")
    with open(path, "wb") as fh:  # what retrieval/index.py writes (index.py:37-40)
        pickle.dump(IndexedCorpus(model.corpus, model.corpus_embeddings.to(torch.float32).cpu()), fh)
    m2 = PremiseRetriever(model.encoder, max_seq_len=g["max_seq_len"])
    m2.load_corpus(path)
    assert not m2.embeddings_staled and m2.corpus_embeddings.device.type == "cpu"
    q = g["queries"][0]
    a = m2.retrieve(q["state"], q["path"], "thm0", Pos(*q["pos"]), 10)
    b = model.retrieve(q["state"], q["path"], "thm0", Pos(*q["pos"]), 10)
    assert [p.full_name for p in a[0]] == [p.full_name for p in b[0]] and a[1] == b[1]
    assert m2.corpus_embeddings.is_cuda and m2.corpus_embeddings.dtype == torch.bfloat16  # moved + cast once


def test_full_size_properties_130k_b256_k100():
    """BASELINE config 2 shape: N=130,000 premises, D=1472, B=256 states, k=100."""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3407)
    rng = np.random.default_rng(3407)
    N, D, B, k, F = 130_000, 1472, 256, 100, 5000
    E = torch.nn.functional.normalize(torch.randn(N, D, generator=gen, device="cuda"), dim=1).to(torch.bfloat16)
    Q = torch.nn.functional.normalize(torch.randn(B, D, generator=gen, device="cuda"), dim=1).to(torch.bfloat16)
    m, acc = hh.synth_masks(rng, N, B, F)
    dm = hh.masks_to_device(m, Q.device)
    ids, sc, cnt = hh.sim_topk(Q, E, k, dm)
    S = (Q.float() @ E.float().T).cpu().numpy()  # plain torch fp32 reference of the similarity GEMM
    hh.check_topk_against_scores(ids.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy(), S, acc, k, tol=2e-5)
    ids2, sc2, cnt2 = hh.sim_topk(Q, E, k, dm, flags=_lib.RP_TOPK_DENSE)
    assert torch.equal(ids, ids2) and torch.equal(sc, sc2) and torch.equal(cnt, cnt2)
    # 8-way row shard + merge == single shot (north_star's multi-GPU layout, on one GPU)
    f, ek, bt, own, qk = dm
    bounds = np.linspace(0, N, 9).astype(int)
    parts = [hh.sim_topk(Q, E[lo:hi].contiguous(), k, (f[lo:hi].contiguous(), ek[lo:hi].contiguous(), bt, own, qk),
                         id_offset=int(lo)) for lo, hi in zip(bounds[:-1], bounds[1:])]
    mi, ms, mc = hh.topk_merge(torch.stack([p[1] for p in parts]), torch.stack([p[0] for p in parts]),
                               torch.stack([p[2] for p in parts]))
    assert torch.equal(mi, ids) and torch.equal(ms, sc) and torch.equal(mc, cnt)


def test_retrieve_from_the_reference_indexed_corpus_pickle(golden_dir):
    """G14: load the index file written by the REFERENCE's retrieval/index.py (pickled reference classes inside) and
    retrieve from it: the premises and scores the reference's own ``retrieve`` returned from that file."""
    import json

    from reprover_amd import synth
    from reprover_amd.common import Pos
    from reprover_amd.retrieval.model import PremiseRetriever

    g = json.load(open(os.path.join(golden_dir, "g14_reference_indexed_corpus.json")))
    cfg = synth.t5_config("tiny")
    model = PremiseRetriever.from_state_dict(cfg, synth.synth_state_dict(cfg, seed=g["weight_seed"]), 512, "cuda:0")
    model.load_corpus(os.path.join(golden_dir, "g14_reference_indexed_corpus.pickle"))
    assert not model.embeddings_staled and len(model.corpus) == g["N"]
    where = {id(p): i for i, p in enumerate(model.corpus.all_premises)}
    for use_graphs in (True, False):
        model.use_graphs = use_graphs
        for j, q in enumerate(g["queries"]):
            prem, scores = model.retrieve(q["state"], q["path"], f"thm{j}", Pos(*q["pos"]), g["k"])
            want = np.array(q["scores"])
            assert np.abs(np.array(scores) - want).max() <= 1e-2  # bf16 index + bf16-operand encode of the state
            got = [where[id(p)] for p in prem]
            for r, (a, b) in enumerate(zip(got, q["ids"])):  # ids wherever the golden gap to both neighbours exceeds 2 tol
                gap = min(abs(want[r] - want[r - 1]) if r else 1.0, abs(want[r] - want[r + 1]) if r + 1 < len(want) else 1.0)
                assert a == b or gap <= 2e-2, (j, r, got, q["ids"])


@pytest.mark.parametrize("index_dtype", ["bf16", "fp8"])
def test_k_beyond_1024_is_served_in_pages(index_dtype):
    """The reference's get_nearest_premises accepts any k (common.py:299-326); one rp_sim_topk call sorts at most 1024
    keys per query.  Larger k goes page by page (rp_sim_topk_after continues the ranking behind the previous page's last
    entry): the concatenation must be exactly the oracle's masked ranking on the same operands - ties included (a dozen
    duplicated rows straddle the page boundaries) - and a query with fewer than k accessible premises raises ValueError."""
    from reprover_amd.common import Fp8Index
    from oracle import fp8_ref

    files = synth.synth_corpus_records(40, 6000, seed=77, max_imports=8)
    path = os.path.join(tempfile.mkdtemp(), "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    corpus, ref = Corpus(path), common_ref.CorpusRef(path)
    N, D, B, k = len(corpus), 128, 6, 2500
    rng = np.random.default_rng(78)
    E = rng.standard_normal((N, D)).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    E[1000:1012] = E[999]  # exact ties: the id order decides, also across a page boundary
    Qm = rng.standard_normal((B, D)).astype(np.float32)
    Qm /= np.linalg.norm(Qm, axis=1, keepdims=True)
    late = [f for f in files[-6:]]  # late files import (transitively) most of the corpus
    ctxs = [Context(f["path"], f"t{j}", Pos(10_000, 0), f"h{j} ⊢ g") for j, f in enumerate(late)]
    rctx = [common_ref.ContextRef(c.path, c.theorem_full_name, common_ref.Pos(*c.theorem_pos), c.state) for c in ctxs]
    n_acc = [int(corpus.accessible_mask(c.path, c.theorem_pos).sum()) for c in ctxs]
    keep = [j for j in range(B) if n_acc[j] >= k]
    assert len(keep) >= 3, n_acc
    Ed, Qd = torch.from_numpy(E).cuda(), torch.from_numpy(Qm).cuda()
    where = {id(p): i for i, p in enumerate(corpus.all_premises)}
    if index_dtype == "bf16":
        Eb, Qb = _bf16_round(E), _bf16_round(Qm)
        operand, tol = Ed, 1e-5
        want_i, want_s = ref.get_nearest_premises(Eb, [rctx[j] for j in keep], Qb[keep], k)
    else:
        operand, tol = Fp8Index.quantize(Ed), 1e-5
        q8 = Fp8Index.quantize(Qd)
        S = fp8_ref.scores_fp8(q8.codes.cpu().numpy(), q8.scale.cpu().numpy(), operand.codes.cpu().numpy(),
                               operand.scale.cpu().numpy())
        acc = np.stack([corpus.accessible_mask(ctxs[j].path, ctxs[j].theorem_pos) for j in keep])
        want_i, want_s = common_ref.masked_topk(S[keep], acc, k)
        want_i, want_s = want_i.tolist(), want_s.tolist()
    prem, scores = corpus.get_nearest_premises(operand, [ctxs[j] for j in keep], Qd[keep], k)
    got = [[where[id(p)] for p in row] for row in prem]
    assert all(len(r) == k and len(set(r)) == k for r in got)
    checked, bad = hh.gap_rule_ids(got, want_i, want_s, tol=2e-6)
    assert bad == 0 and checked > k
    assert np.abs(np.array(scores) - np.array(want_s)).max() < tol
    for row, sc in zip(got, scores):  # one total order across the pages: score descending, id ascending among equals
        assert all(sc[i] > sc[i + 1] or (sc[i] == sc[i + 1] and row[i] < row[i + 1]) for i in range(k - 1))
    assert n_acc[0] < N  # (a theorem never sees the premises behind it in its own file)
    with pytest.raises(ValueError):
        corpus.get_nearest_premises(operand, [ctxs[0]], Qd[[0]], N)


@pytest.mark.parametrize("index_dtype", ["bf16", "fp8"])
def test_paging_with_queries_that_run_out_of_premises(index_dtype):
    """Later pages when some queries are already exhausted (bound (-inf, INT_MAX): the second-generation filter's PAGED
    pre-test `score <= after_score` admits none of their rows) and others come back with a short last page:
    130 theorems spread over the corpus's files, from ones that see a few hundred premises to ones that see nearly all 20 k.
    Every query's list is its whole masked ranking up to k, in (score desc, id asc) order; counts = min(k, accessible)."""
    from reprover_amd.common import Fp8Index
    from oracle import fp8_ref

    files = synth.synth_corpus_records(60, 20000, seed=179, max_imports=10)
    path = os.path.join(tempfile.mkdtemp(), "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    corpus = Corpus(path)
    N, D, k, B = len(corpus), 1472, 2300, 130
    rng = np.random.default_rng(180)
    E = rng.standard_normal((N, D)).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    Qm = rng.standard_normal((B, D)).astype(np.float32)
    Qm /= np.linalg.norm(Qm, axis=1, keepdims=True)
    ctxs = [Context(files[(j * 7) % len(files)]["path"], f"t{j}", Pos(10_000, 0), f"h{j} ⊢ g") for j in range(B)]
    acc = np.stack([corpus.accessible_mask(c.path, c.theorem_pos) for c in ctxs])
    n_acc = acc.sum(1)
    assert (n_acc < 1024).any() and ((n_acc > 1024) & (n_acc < k)).any() and (n_acc >= k).any(), sorted(n_acc.tolist())[::13]
    Ed, Qd = torch.from_numpy(E).cuda(), torch.from_numpy(Qm).cuda()
    if index_dtype == "bf16":
        operand = Ed
        S = (_bf16_round(Qm).astype(np.float64) @ _bf16_round(E).astype(np.float64).T).astype(np.float32)
        tol = 1e-6
    else:
        operand = Fp8Index.quantize(Ed)
        q8 = Fp8Index.quantize(Qd)
        S = fp8_ref.scores_fp8(q8.codes.cpu().numpy(), q8.scale.cpu().numpy(), operand.codes.cpu().numpy(),
                               operand.scale.cpu().numpy())
        tol = 4e-6
    ids, scores, counts = corpus.nearest_premise_ids(operand, ctxs, Qd, k)
    ids, scores, counts = ids.cpu().numpy(), scores.cpu().numpy(), counts.cpu().numpy()
    assert (counts == np.minimum(k, n_acc)).all()
    want_i, want_s = [], []
    for j in range(B):
        n = int(counts[j])
        row, sc = ids[j, :n].tolist(), scores[j, :n]
        assert len(set(row)) == n and acc[j][row].all()
        assert all(sc[i] > sc[i + 1] or (sc[i] == sc[i + 1] and row[i] < row[i + 1]) for i in range(n - 1)), j
        cand = np.flatnonzero(acc[j])
        order = cand[np.lexsort((cand, -S[j, cand]))][:n]
        assert np.abs(sc - S[j, order]).max() < 1e-5
        if n < k:  # an exhausted query returned EVERY accessible premise
            assert set(row) == set(cand.tolist())
        want_i.append(order.tolist())
        want_s.append(S[j, order].tolist())
    checked, bad = hh.gap_rule_ids([ids[j, : int(counts[j])].tolist() for j in range(B)], want_i, want_s, tol=tol)
    assert bad == 0 and checked > 1000


@pytest.mark.parametrize("index_dtype", ["bf16", "fp8"])
@pytest.mark.parametrize("B", [6, 130])
def test_paging_through_the_two_pass_plan(index_dtype, B):
    """k > 1024 on an index of more than 16,384 rows: every page runs the TWO-PASS plan (sample scan -> bound -> filter ->
    gather / select) with the `after` bound of the page before - the first-generation filter at B <= 128, the
    second-generation one (private survivor runs, D % 64 == 0) at B > 128, bf16 and e4m3 rows of d_model 1472 (11.5
    k-tiles of e4m3: the half-tile form).  The concatenated pages must be exactly the oracle's masked ranking on the same
    operands, exact ties across a page boundary included (the small-index paging test only reaches the dense
    plan)."""
    from reprover_amd.common import Fp8Index
    from oracle import fp8_ref

    files = synth.synth_corpus_records(60, 20000, seed=177, max_imports=10)
    path = os.path.join(tempfile.mkdtemp(), "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    corpus = Corpus(path)
    N, D, k = len(corpus), 1472, 2300
    assert N > 16384
    rng = np.random.default_rng(178 + B)
    E = rng.standard_normal((N, D)).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    Qm = rng.standard_normal((B, D)).astype(np.float32)
    Qm /= np.linalg.norm(Qm, axis=1, keepdims=True)
    late = files[-8:]
    ctxs = [Context(late[j % 8]["path"], f"t{j}", Pos(10_000, 0), f"h{j} ⊢ g") for j in range(B)]
    acc = np.stack([corpus.accessible_mask(c.path, c.theorem_pos) for c in ctxs])
    assert (acc.sum(1) >= k).all(), acc.sum(1).min()
    Ed, Qd = torch.from_numpy(E).cuda(), torch.from_numpy(Qm).cuda()
    if index_dtype == "bf16":
        Eb, Qb = _bf16_round(E), _bf16_round(Qm)
        # exact ties straddling the first page boundary of query 0: twelve copies of its 1020th-best accessible row
        S0 = Qb[0] @ Eb.T
        order = np.argsort(-np.where(acc[0], S0, -np.inf), kind="stable")
        dup = [int(i) for i in order[1019:1031]]
        E[dup] = E[dup[0]]
        Ed = torch.from_numpy(E).cuda()
        Eb = _bf16_round(E)
        operand = Ed
        S = Qb.astype(np.float64) @ Eb.astype(np.float64).T
        S = S.astype(np.float32)
    else:
        operand = Fp8Index.quantize(Ed)
        q8 = Fp8Index.quantize(Qd)
        S = fp8_ref.scores_fp8(q8.codes.cpu().numpy(), q8.scale.cpu().numpy(), operand.codes.cpu().numpy(),
                               operand.scale.cpu().numpy())
    want_i, want_s = common_ref.masked_topk(S, acc, k)
    ids, scores, counts = corpus.nearest_premise_ids(operand, ctxs, Qd, k)
    ids, scores, counts = ids.cpu().numpy(), scores.cpu().numpy(), counts.cpu().numpy()
    assert (counts == k).all()
    for j in range(B):
        row = ids[j].tolist()
        assert len(set(row)) == k, f"query {j}: a row was returned on two pages (or dropped)"
        assert acc[j][row].all()
        sc = scores[j]
        assert all(sc[i] > sc[i + 1] or (sc[i] == sc[i + 1] and row[i] < row[i + 1]) for i in range(k - 1)), j
    # (the e4m3 scan's scores lie within 2e-6 of the exact value of the quantised dot product, DESIGN.md section 2: its
    # ranks are compared where the oracle's gap exceeds twice that and more)
    checked, bad = hh.gap_rule_ids(ids.tolist(), want_i.tolist(), want_s.tolist(), tol=1e-6 if index_dtype == "bf16" else 4e-6)
    assert bad == 0 and checked > B * k // (20 if index_dtype == "bf16" else 100)
    assert np.abs(scores - want_s).max() < 1e-5
    if index_dtype == "bf16":  # the twelve tied rows of query 0 come out in id order, across the page boundary
        pos = [ids[0].tolist().index(i) for i in sorted(dup)]
        assert pos == list(range(pos[0], pos[0] + 12)) and pos[0] <= 1023 <= pos[-1] + 12
