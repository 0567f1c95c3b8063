"""Generate tests/golden/* by running the REFERENCE itself (lean-dojo/ReProver, imported from
/root/reference through tests/golden/ref_harness.py) and HuggingFace transformers on CPU.

Authoring container only.  Only the resulting small data files are committed; no reference
source travels.  Usage:  python tests/golden/make_golden.py [g1 .. g9]   (default: all)

Every fixture records the inputs and the reference's outputs; while generating, the oracle
(oracle/) is checked against the reference so a drifting restatement fails here first.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import time

if os.environ.get("PYTHONHASHSEED") != "0":
    # Reproducible bytes: the reference iterates over Python sets of strings (the order of a tactic's positive premises,
    # networkx's successor lists inside the pickled Corpus of G14), so the interpreter's string-hash seed is pinned.
    os.environ["PYTHONHASHSEED"] = "0"
    os.execv(sys.executable, [sys.executable] + sys.argv)

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))  # tests/golden/ -> repo root
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_harness as H  # noqa: E402

common, rm = H.install()
from transformers import ByT5Tokenizer  # noqa: E402
from transformers.models.t5.modeling_t5 import T5Attention  # noqa: E402

from oracle import common_ref, t5_ref  # noqa: E402
from reprover_amd import synth  # noqa: E402

OUT = HERE
os.makedirs(OUT, exist_ok=True)


def hf_cfg(cfg):
    return dict(
        vocab_size=cfg["vocab_size"],
        d_model=cfg["d_model"],
        d_kv=cfg["d_kv"],
        d_ff=cfg["d_ff"],
        num_layers=cfg["num_layers"],
        num_decoder_layers=1,
        num_heads=cfg["num_heads"],
        feed_forward_proj="gated-gelu",
        tie_word_embeddings=False,
    )


# ----------------------------------------------------------------------------------------------
def g1_tokenizer():
    tok = ByT5Tokenizer()
    rng = np.random.default_rng(1)
    texts = [
        "abc",
        "",
        " ",
        "theorem foo (n : ℕ) : n + 0 = n := by simp",
        "ℕ ⊢ «x» → ∀ y, y ≤ x",
        "a\n\nb  c\t\td",
        "a </s> b",
        "x<pad>y",
        "<unk>",
        "<extra_id_0> z <extra_id_124>",
        "<extra_id_125> not special",
        "tail eos </s>",
        "</s>",
        "<a>Nat.add_comm</a> : ∀ (n m : ℕ), n + m = m + n",
        "ℕℕℕℕℕℕ",  # multi-byte char split by truncation at max_length 16/8
        "x" * 15,
        "x" * 16,
        "x" * 17,
        synth.synth_text(rng, 1023),
        synth.synth_text(rng, 1024),
        synth.synth_text(rng, 2047),
        synth.synth_text(rng, 2048),
        synth.synth_text(rng, 3000),
        "𝔸𝔹ℂ 😀 日本語",
    ]
    cases = []
    for ml in (8, 16, 1024, 2048):
        enc = tok(texts, padding="longest", max_length=ml, truncation=True, return_tensors="np")
        ids, mask = enc["input_ids"], enc["attention_mask"]
        o_ids, o_mask = t5_ref.byt5_batch(texts, ml)
        assert np.array_equal(ids, o_ids) and np.array_equal(mask, o_mask), f"oracle tokenizer drift @ {ml}"
        rows = [ids[i, : int(mask[i].sum())].tolist() for i in range(len(texts))]
        cases.append({"max_length": ml, "padded_len": int(ids.shape[1]), "ids": rows})
    json.dump({"texts": texts, "cases": cases}, open(os.path.join(OUT, "g1_tokenizer.json"), "w"), ensure_ascii=False)
    print("g1 ok:", len(texts), "texts x", len(cases), "max_lengths")


# ----------------------------------------------------------------------------------------------
def g2_serialize():
    P = H.Pos
    recs = [
        ("A.lean", "Nat.add_comm", "theorem Nat.add_comm (n m : ℕ) : n + m = m + n := sorry"),
        ("A.lean", "Nat.add_comm", "theorem add_comm (n m : ℕ) : n + m = m + n"),
        ("A.lean", "Nat.add_comm", "theorem _root_.Nat.add_comm : True"),
        ("A.lean", "Foo.bar.baz", "lemma baz : 1 = 1\nlemma bar.baz' : 2 = 2"),
        ("A.lean", "Foo.bar.baz", "lemma «baz» : 1 = 1"),
        ("A.lean", "Foo.bar.baz", "lemma «bar.baz» «baz» : 1 = 1"),
        ("A.lean", "Foo.bar.baz", "no match here"),
        ("A.lean", "Foo.bar.baz", "baz at start is not preceded by whitespace"),
        ("A.lean", "a.b", "theorem aXb : True  -- '.' is a regex wildcard in the reference"),
        ("A.lean", "List.map", "def map (f : α → β) : List α → List β\n  | [] => []\n  | a::as => f a :: map f as"),
        ("A.lean", "instInhabitedNat", "instance : Inhabited ℕ := ⟨0⟩"),
        ("A.lean", "Set.mem_def", "theorem\tmem_def {a : α} {s : Set α} : a ∈ s ↔ s a"),
        ("A.lean", "x", "def x := x + x"),
        ("A.lean", "Real.sqrt", "noncomputable def Real.sqrt (x : ℝ) : ℝ := NNReal.sqrt (Real.toNNReal x)"),
        ("A.lean", "Nat.succ_le", "theorem succ_le {n m : ℕ} : succ n ≤ m ↔ n < m  -- succ_le succ_le"),
        ("A.lean", "Foo.Bar", "structure Bar where\n  x : ℕ\n\n«Foo.Bar» Bar"),
    ]
    cases = []
    for path, name, code in recs:
        ref = common.Premise(path, name, P(1, 0), P(2, 0), code).serialize()
        mine = common_ref.PremiseRef(path, name, common_ref.Pos(1, 0), common_ref.Pos(2, 0), code).serialize()
        assert ref == mine, (name, code, ref, mine)
        cases.append({"path": path, "full_name": name, "code": code, "serialized": ref})
    # File.from_data filters + a full synthetic corpus' serialisations
    files = synth.synth_corpus_records(12, 150, seed=7)
    kept = []
    for fd in files:
        f = common.File.from_data(fd)
        mine = common_ref.premises_of_file(fd)
        assert [p.full_name for p in f.premises] == [p.full_name for p in mine]
        assert [p.serialize() for p in f.premises] == [p.serialize() for p in mine]
        kept.append({"path": fd["path"], "names": [p.full_name for p in f.premises],
                     "serialized": [p.serialize() for p in f.premises]})
    json.dump({"cases": cases, "corpus_seed": 7, "corpus_files": 12, "corpus_premises": 150, "kept": kept},
              open(os.path.join(OUT, "g2_serialize.json"), "w"), ensure_ascii=False)
    print("g2 ok:", len(cases), "premises;", sum(len(k["names"]) for k in kept), "kept of synthetic corpus")


# ----------------------------------------------------------------------------------------------
def g3_buckets():
    d = torch.arange(-2200, 2201, dtype=torch.long)
    b = T5Attention._relative_position_bucket(d, bidirectional=True, num_buckets=32, max_distance=128).numpy()
    mine = t5_ref.relative_position_bucket(d.numpy(), 32, 128)
    assert np.array_equal(b, mine)
    np.savez_compressed(os.path.join(OUT, "g3_buckets.npz"), rel=d.numpy().astype(np.int32), bucket=b.astype(np.int8))
    print("g3 ok: buckets for", len(d), "offsets; distinct", len(np.unique(b)))


# ----------------------------------------------------------------------------------------------
def _texts_with_lengths(rng, lens):
    return [synth.synth_text(rng, int(n) - 1) for n in lens]  # +EOS = n tokens


def _ref_encode(model, texts, max_len, bs):
    """The reference's own tokenise → _encode, in corpus-order batches (model.py:197-208)."""
    outs = []
    for i in range(0, len(texts), bs):
        t = model.tokenizer(texts[i : i + bs], padding="longest", max_length=max_len, truncation=True,
                            return_tensors="pt")
        with torch.no_grad():
            outs.append(model._encode(t.input_ids, t.attention_mask))
    return torch.cat(outs)


def g4_tiny():
    cfg = synth.t5_config("tiny")
    sd = synth.synth_state_dict(cfg)
    model = H.offline_retriever(rm, hf_cfg(cfg), sd, max_seq_len=512)
    assert model.encoder.dtype == torch.float32
    rng = np.random.default_rng(4)
    lens = [2, 3, 9, 31, 32, 33, 64, 65, 100, 127, 128, 129, 200, 257, 300, 512]
    texts = _texts_with_lengths(rng, lens)
    texts[3] = "n : ℕ ⊢ n + 0 = n"
    t = model.tokenizer(texts, padding="longest", max_length=512, truncation=True, return_tensors="pt")
    with torch.no_grad():
        hidden = model.encoder(input_ids=t.input_ids, attention_mask=t.attention_mask).last_hidden_state
        emb = model._encode(t.input_ids, t.attention_mask)
    o_hidden = t5_ref.encoder_forward(cfg, sd, t.input_ids.numpy(), t.attention_mask.numpy())
    o_emb = t5_ref.encode(cfg, sd, t.input_ids.numpy(), t.attention_mask.numpy())
    m = t.attention_mask.bool()
    dh = (hidden - o_hidden)[m].abs().max().item()
    de = (emb - o_emb).abs().max().item()
    print(f"g4: oracle vs HF  max|Δhidden| = {dh:.2e}  max|Δemb| = {de:.2e}")
    assert dh < 5e-5 and de < 1e-6
    # padding invariance: each text alone == inside the batch
    solo = _ref_encode(model, texts, 512, 1)
    print(f"g4: batch vs solo max|Δemb| = {(solo - emb).abs().max().item():.2e}")
    np.savez_compressed(
        os.path.join(OUT, "g4_tiny.npz"),
        texts=np.array(texts, dtype=object),
        input_ids=t.input_ids.numpy().astype(np.int32),
        attention_mask=t.attention_mask.numpy().astype(np.int8),
        hidden_last_rows=hidden[torch.arange(len(texts)), t.attention_mask.sum(1) - 1].numpy(),
        hidden_first_rows=hidden[:, 0].numpy(),
        emb=emb.numpy(),
        seed=np.int64(synth.SEED),
    )
    print("g4 ok")


def g5_small(scale="sharp"):
    """scale="sharp": fixture G5 (the stress family of synth.synth_state_dict); scale="hf": fixture G5h - the same 16
    texts on weights at exactly HF's init scales, SURVEY.md section 8c's recipe, on which the written contract holds."""
    cfg = synth.t5_config("byt5-small")
    t0 = time.time()
    sd = synth.synth_state_dict(cfg, scale=scale)
    print(f"g5[{scale}]: weights generated in {time.time() - t0:.1f}s")
    model = H.offline_retriever(rm, hf_cfg(cfg), sd, max_seq_len=2048)
    n_par = sum(p.numel() for p in model.encoder.parameters())
    print("g5: encoder params", n_par)
    rng = np.random.default_rng(5)
    lens = [8, 17, 33, 64, 100, 128, 180, 256, 300, 400, 512, 700, 1024, 1500, 2048, 2600]
    texts = _texts_with_lengths(rng, lens)
    texts[4] = "α : Type u_1\ninst✝ : LinearOrder α\na b : α\nh : a ≤ b\n⊢ max a b = b" + synth.synth_text(rng, 20)
    t0 = time.time()
    emb = _ref_encode(model, texts, 2048, 4)
    print(f"g5: HF fp32 encode of {sum(min(l, 2048) for l in lens)} tokens in {time.time() - t0:.1f}s")
    t0 = time.time()
    o_emb = t5_ref.encode_texts(cfg, sd, texts, 2048, 4)
    de = (emb - o_emb).abs().max().item()
    print(f"g5: oracle vs HF max|Δemb| = {de:.2e}  ({time.time() - t0:.1f}s)")
    assert de < 2e-5
    # the reference's GPU numerics (bf16 everywhere) for the tolerance envelope
    t0 = time.time()
    model_bf = model.to(torch.bfloat16)
    emb_bf = _ref_encode(model_bf, texts, 2048, 4).float()
    print(f"g5: HF bf16 vs fp32 max|Δemb| = {(emb_bf - emb).abs().max().item():.2e}; "
          f"min cos = {torch.nn.functional.cosine_similarity(emb_bf, emb).min().item():.5f} ({time.time() - t0:.1f}s)")
    np.savez_compressed(
        os.path.join(OUT, "g5_byt5_small.npz" if scale == "sharp" else "g5h_byt5_small.npz"),
        texts=np.array(texts, dtype=object),
        emb=emb.numpy(),
        emb_hf_bf16=emb_bf.numpy().astype(np.float16),
        seed=np.int64(synth.SEED),
        **({} if scale == "sharp" else {"weight_scale": np.array(scale)}),
    )
    print(f"g5[{scale}] ok")


# ----------------------------------------------------------------------------------------------
def _ref_nearest_indexes(corpus, E, ctxs, Q, k):
    """Reference get_nearest_premises → (indexes, scores); Premise objects are mapped back to
    their position in all_premises by object identity."""
    where = {id(p): i for i, p in enumerate(corpus.all_premises)}
    prem, scores = corpus.get_nearest_premises(E, ctxs, Q, k)
    return [[where[id(p)] for p in row] for row in prem], scores


def g6_nearest():
    files = synth.synth_corpus_records(40, 1000, seed=6)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "corpus.jsonl")
        synth.write_corpus_jsonl(path, files)
        corpus = common.Corpus(path)
        ocorpus = common_ref.CorpusRef(path)
    N = len(corpus)
    assert N == len(ocorpus)
    rng = np.random.default_rng(66)
    D = 64
    E = rng.standard_normal((N, D)).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    B = 24
    Q = rng.standard_normal((B, D)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    # query positions: late files so that plenty is accessible, plus a few early/edge ones
    qfiles = list(rng.integers(25, 40, size=B - 4)) + [39, 30, 1, 0]
    ctxs, octxs, qmeta = [], [], []
    for j, f in enumerate(qfiles):
        fpath = files[int(f)]["path"]
        prem = corpus.get_premises(fpath)
        if prem and j % 3 != 0:
            anchor = prem[int(rng.integers(len(prem)))]
            pos = (anchor.end.line_nb, anchor.end.column_nb + int(rng.integers(0, 2)))
        else:
            pos = (int(rng.integers(1, 400)), int(rng.integers(0, 50)))
        ctxs.append(common.Context(fpath, f"thm{j}", H.Pos(*pos), f"x{j} ⊢ y"))
        octxs.append(common_ref.ContextRef(fpath, f"thm{j}", common_ref.Pos(*pos), f"x{j} ⊢ y"))
        qmeta.append({"path": fpath, "pos": list(pos)})
    # accessibility: reference set-form vs oracle, and index-form
    acc = np.zeros((B, N), dtype=bool)
    for j, c in enumerate(ctxs):
        s = corpus.get_accessible_premises(c.path, c.theorem_pos)
        row = np.array([p in s for p in corpus.all_premises])
        keys = ocorpus.accessible_keys(c.path, octxs[j].theorem_pos)
        orow = np.array([(p.path, p.full_name) in keys for p in ocorpus.all_premises])
        assert np.array_equal(row, orow)
        acc[j] = row
    out = {"acc_counts": acc.sum(1).tolist()}
    results = {}
    Et, Qt = torch.from_numpy(E), torch.from_numpy(Q)
    for k in (1, 10, 100):
        ok = [j for j in range(B) if acc[j].sum() >= k]
        idx, sc = _ref_nearest_indexes(corpus, Et, [ctxs[j] for j in ok], Qt[ok], k)
        oidx, osc = ocorpus.get_nearest_premises(E, [octxs[j] for j in ok], Q[ok], k)
        assert idx == oidx, f"oracle nearest drift at k={k}"
        assert np.allclose(np.array(sc), np.array(osc), atol=1e-6)
        results[str(k)] = {"queries": ok, "ids": idx, "scores": sc}
    # the ValueError case: a query with < k accessible premises (common.py:323-324)
    bad = [j for j in range(B) if acc[j].sum() < 100]
    assert bad, "need at least one query with <100 accessible premises"
    try:
        corpus.get_nearest_premises(Et, [ctxs[bad[0]]], Qt[bad[:1]], 100)
        raise AssertionError("reference did not raise")
    except ValueError:
        pass
    try:
        ocorpus.get_nearest_premises(E, [octxs[bad[0]]], Q[bad[:1]], 100)
        raise AssertionError("oracle did not raise")
    except ValueError:
        pass
    out.update({"corpus_seed": 6, "n_files": 40, "n_premises": 1000, "N": N, "D": D, "emb_seed": 66,
                "queries": qmeta, "results": results, "value_error_query": bad[0]})
    json.dump(out, open(os.path.join(OUT, "g6_nearest.json"), "w"))
    np.savez_compressed(os.path.join(OUT, "g6_nearest.npz"), E=E, Q=Q, acc=np.packbits(acc, axis=1))
    print("g6 ok: N =", N, "accessible counts min/max", acc.sum(1).min(), acc.sum(1).max())


# ----------------------------------------------------------------------------------------------
def _g7_run_predict(model, ctxs, where, max_len):
    """The reference's predict_step (model.py:281-327) over the states in eval batches of 64 (retrieval/confs/*.yaml);
    returns (ids, scores) per state."""
    ids_all, sc_all = [], []
    for i in range(0, len(ctxs), 64):
        batch = ctxs[i : i + 64]
        tok = model.tokenizer([c.serialize() for c in batch], padding="longest", max_length=max_len,
                              truncation=True, return_tensors="pt")  # datamodule.py:130-144
        model.predict_step_outputs = []
        b = {"context": batch, "context_ids": tok.input_ids, "context_mask": tok.attention_mask}
        for key in ("url", "commit", "file_path", "full_name", "start", "tactic_idx", "all_pos_premises"):
            b[key] = [None] * len(batch)
        with torch.no_grad():
            model.predict_step(b, 0)
        for rec in model.predict_step_outputs:
            ids_all.append([where[id(p)] for p in rec["retrieved_premises"]])
            sc_all.append([float(x) for x in rec["scores"]])
    return ids_all, sc_all


def g7_predict(scale="sharp"):
    """BASELINE config 1: 1k synthetic premises, 128 states, top-10 through the reference's
    reindex_corpus + predict_step logic (Lightning is stubbed; the hooks' bodies are the
    reference's).

    scale="sharp": fixture G7 (stress-family weights, independent random premise bodies).
    scale="hf": fixture **G7h** - weights at HF's init scales (the family the written contract was derived on) and the
    FAMILY corpus of synth.synth_family_corpus_records, 15 of every 16 states built from a family's base text:
    the fp32 scores of a state's top-10 then step down by several 1e-2 per rank, so the id comparison ("equal wherever
    the oracle's gap to both neighbours exceeds 2 x tol") has hundreds of ranks to check instead of a few dozen."""
    hf = scale == "hf"
    tag = "g7h" if hf else "g7"
    cfg = synth.t5_config("byt5-small")
    sd = synth.synth_state_dict(cfg, scale=scale)
    model = H.offline_retriever(rm, hf_cfg(cfg), sd, max_seq_len=1024)
    corpus_seed = 171 if hf else 71
    if hf:
        files, families = synth.synth_family_corpus_records(60, 1000, seed=corpus_seed, code_bytes=(24, 96))
    else:
        files, families = synth.synth_corpus_records(60, 1000, seed=corpus_seed, code_bytes=(24, 96)), []
    td = tempfile.mkdtemp()
    path = os.path.join(td, "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    model.load_corpus(path)
    N = len(model.corpus)
    t0 = time.time()
    model.reindex_corpus(batch_size=64)
    print(f"{tag}: reference reindex_corpus of {N} premises took {time.time() - t0:.1f}s")
    E = model.corpus_embeddings
    rng = np.random.default_rng(172 if hf else 72)
    B = 128
    ctxs, qmeta = [], []
    for j in range(B):
        while True:  # the reference raises ValueError when < k premises are accessible
            f = int(rng.integers(30, 60))
            pos = (int(rng.integers(1, 300)), int(rng.integers(0, 40)))
            acc = model.corpus.get_accessible_premises(files[f]["path"], H.Pos(*pos))
            if len(acc) >= 12:
                break
        state = synth.synth_state(rng, int(rng.integers(40, 200)))
        if hf and j % 16 != 15:
            # a family all of whose members this state may use (own file before pos, or an imported file)
            names = {(p.path, p.full_name) for p in acc}
            ok = [g for g in families if all((files[g["file"]]["path"], n) in names for n in g["members"])
                  and len(g["members"]) >= 10]
            if ok:
                base = ok[int(rng.integers(len(ok)))]["base"]
                base = synth.corrupt_text(rng, base, 0.04)
                h = int(rng.integers(len(base) // 4, 3 * len(base) // 4))
                state = base[:h] + " ⊢" + base[h:]
        ctxs.append(common.Context(files[f]["path"], f"thm{j}", H.Pos(*pos), state))
        qmeta.append({"path": files[f]["path"], "pos": list(pos), "state": state})
    where = {id(p): i for i, p in enumerate(model.corpus.all_premises)}
    model.num_retrieved = 10  # BASELINE config 1: top-10 (the hook reads self.num_retrieved, model.py:288)
    t0 = time.time()
    ids_all, sc_all = _g7_run_predict(model, ctxs, where, 1024)
    print(f"{tag}: reference predict of {B} states took {time.time() - t0:.1f}s")
    from oracle.parity_margins import gap_rule_ids
    checked, _ = gap_rule_ids(ids_all, ids_all, sc_all, tol=1e-2)
    print(f"{tag}: ranks the gap rule (2 x 1e-2 to both neighbours) can check: {checked} of {B * 10}")
    # single-query retrieve() (model.py:338-375) on the first 4 states
    single = []
    for j in range(4):
        c = ctxs[j]
        prem, sc = model.retrieve(c.state, c.path, c.theorem_full_name, c.theorem_pos, 10)
        single.append({"ids": [where[id(p)] for p in prem], "scores": sc})
        assert single[-1]["ids"] == ids_all[j]
    # The reference's own GPU numerics (model.py:59-64: everything bf16) on the same inputs: how far
    # HuggingFace-bf16 lands from its fp32 self, measured at the golden ids (the tolerance envelope), and the ids its
    # own predict_step returns (bf16 embeddings, bf16 similarity matrix: common.py:307-308 as the GPU mode runs it).
    t0 = time.time()
    E32 = E.clone()
    model_bf = model.to(torch.bfloat16)
    model_bf.embeddings_staled = True
    model_bf.reindex_corpus(batch_size=64)
    Ebf = model_bf.corpus_embeddings.float()
    hf_bf16_scores = []
    for i in range(0, B, 64):
        batch = ctxs[i : i + 64]
        tok = model_bf.tokenizer([c.serialize() for c in batch], padding="longest", max_length=1024,
                                 truncation=True, return_tensors="pt")
        with torch.no_grad():
            qbf = model_bf._encode(tok.input_ids, tok.attention_mask).float()
        for j in range(len(batch)):
            hf_bf16_scores.append((Ebf[ids_all[i + j]] @ qbf[j]).tolist())
    hf_ids, hf_sc = _g7_run_predict(model_bf, ctxs, where, 1024)
    _, hf_bad = gap_rule_ids(hf_ids, ids_all, sc_all, tol=1e-2)
    hf_top1 = float(np.mean([a[0] == b[0] for a, b in zip(hf_ids, ids_all)]))
    hf_top10 = float(np.mean([len(set(a) & set(b)) / 10 for a, b in zip(hf_ids, ids_all)]))
    d_hf = np.abs(np.array(hf_bf16_scores) - np.array(sc_all))
    cos_hf = torch.nn.functional.cosine_similarity(Ebf, E32, dim=1)
    print(f"{tag}: HF-bf16 vs fp32: max|Δscore| at golden ids {d_hf.max():.3e}, mean {d_hf.mean():.3e}; "
          f"min embedding cosine {cos_hf.min().item():.5f}; its own predict: top-1 agreement {hf_top1:.3f}, top-10 overlap "
          f"{hf_top10:.3f}, gap-rule mismatches {hf_bad} ({time.time() - t0:.0f}s)")
    E = E32
    probe = np.random.default_rng(73).standard_normal((E.shape[1], 4)).astype(np.float32)
    doc = {"corpus_seed": corpus_seed, "n_files": 60, "n_premises": 1000, "code_bytes": [24, 96], "N": N,
           "queries": qmeta, "k": 10, "ids": ids_all, "scores": sc_all, "retrieve": single,
           "max_seq_len": 1024, "batch_size": 64, "hf_bf16_scores_at_gold_ids": hf_bf16_scores,
           "hf_bf16_min_embedding_cosine": float(cos_hf.min()),
           "hf_bf16_predict_ids": hf_ids, "hf_bf16_predict_scores": hf_sc,
           "gap_rule_ranks_checkable": int(checked)}
    if hf:
        doc.update({"weight_scale": "hf", "corpus": "family", "family_size": 12})
    json.dump(doc, open(os.path.join(OUT, f"{tag}_predict.json"), "w"), ensure_ascii=False)
    np.savez_compressed(os.path.join(OUT, f"{tag}_predict.npz"), E_probe=(E @ torch.from_numpy(probe)).numpy(),
                        probe_seed=np.int64(73), E_head=E[:16].numpy(), E_all_f16=E.numpy().astype(np.float16))
    print(f"{tag} ok")


# ----------------------------------------------------------------------------------------------
def g8_eval_data():
    """Evaluation-side host logic: the reference's RetrievalDataset (is_train=False) examples and
    collate, and retrieval/evaluate.py::_eval on a synthetic predictions list."""
    import importlib
    from oracle import eval_ref

    dmod = importlib.import_module("retrieval.datamodule")
    emod = importlib.import_module("retrieval.evaluate")
    files = synth.synth_corpus_records(30, 500, seed=81, max_imports=5)
    td = tempfile.mkdtemp()
    cpath = os.path.join(td, "corpus.jsonl")
    synth.write_corpus_jsonl(cpath, files)
    split = synth.synth_split(files, 40, seed=82, min_file=8)
    spath = os.path.join(td, "val.json")
    json.dump(split, open(spath, "w"))
    corpus = common.Corpus(cpath)
    ocorpus = common_ref.CorpusRef(cpath)
    tok = ByT5Tokenizer()
    ds = dmod.RetrievalDataset([spath], corpus, 3, 1, 256, tok, is_train=False)
    where = {id(p): i for i, p in enumerate(corpus.all_premises)}
    examples = []
    for ex in ds.data:
        examples.append({"file_path": ex["file_path"], "full_name": ex["full_name"], "start": ex["start"],
                         "tactic_idx": ex["tactic_idx"], "state": ex["context"].state,
                         "all_pos_premises": sorted(where[id(p)] for p in ex["all_pos_premises"])})
    mine = eval_ref.load_eval_examples(spath, ocorpus)
    assert mine == examples, "oracle eval-example drift"
    batch = ds.collate(ds.data[:7])
    assert set(batch) == {"context", "context_ids", "context_mask", "url", "commit", "file_path", "full_name",
                          "start", "tactic_idx", "all_pos_premises"}
    # synthetic predictions: 20 distinct premises per example, positives sprinkled in
    rng = np.random.default_rng(83)
    N = len(corpus)
    retrieved, preds = [], []
    for ex_ref, ex in zip(ds.data, examples):
        got = [int(x) for x in rng.choice(N, size=20, replace=False)]
        for p in ex["all_pos_premises"]:
            if rng.random() < 0.6 and p not in got:
                got[int(rng.integers(0, 20))] = p
        assert len(set(got)) == 20
        retrieved.append(got)
        preds.append({"file_path": ex["file_path"], "full_name": ex["full_name"], "start": ex["start"],
                      "tactic_idx": ex["tactic_idx"], "all_pos_premises": ex_ref["all_pos_premises"],
                      "retrieved_premises": [corpus.all_premises[i] for i in got]})
    preds_map = {(p["file_path"], p["full_name"], tuple(p["start"]), p["tactic_idx"]): p for p in preds}
    assert len(preds_map) == len(preds)
    R1, R10, MRR = emod._eval(split, preds_map)
    o = eval_ref.eval_predictions(examples, retrieved)
    assert np.allclose(o, (R1, R10, MRR), atol=1e-9), (o, R1, R10, MRR)
    rec, mrr = eval_ref.validation_metrics([e["all_pos_premises"] for e in examples], retrieved, 20)
    assert abs(rec[0] - R1) < 1e-9 and abs(rec[9] - R10) < 1e-9 and abs(mrr - MRR) < 1e-12  # the two reference formulas agree
    json.dump({"corpus_seed": 81, "n_files": 30, "n_premises": 500, "max_imports": 5, "split_seed": 82,
               "n_theorems": 40, "min_file": 8, "examples": examples, "retrieved": retrieved,
               "R1": R1, "R10": R10, "MRR": MRR, "recall_at_k": rec,
               "collate_keys": sorted(batch), "collate_ids_first7": batch["context_ids"].tolist()},
              open(os.path.join(OUT, "g8_eval.json"), "w"), ensure_ascii=False)
    print(f"g8 ok: {len(examples)} examples, {sum(1 for e in examples if e['all_pos_premises'])} with premises; "
          f"R@1 {R1:.3f} R@10 {R10:.3f} MRR {MRR:.4f}")


# ----------------------------------------------------------------------------------------------
def g9_base_full_depth(scale="sharp"):
    """(scale="hf": fixture G9h, the same texts on HF-init-scale weights.)
    BASELINE configs[4]'s encoder at FULL depth: ByT5-base geometry (d_model 1536, 12 heads, d_ff 3968,
    18 layers) through the reference's tokenise -> _encode with HuggingFace fp32, plus HF-bf16 (the
    reference's GPU numerics) for the tolerance envelope.  8 short texts keep the fp32 CPU run short."""
    cfg = synth.t5_config("byt5-base")
    t0 = time.time()
    sd = synth.synth_state_dict(cfg, scale=scale)
    print(f"g9[{scale}]: weights generated in {time.time() - t0:.1f}s")
    model = H.offline_retriever(rm, hf_cfg(cfg), sd, max_seq_len=1024)
    assert model.encoder.config.num_layers == 18
    rng = np.random.default_rng(9)
    lens = [6, 23, 64, 97, 129, 200, 333, 520]
    texts = _texts_with_lengths(rng, lens)
    texts[2] = "n m : ℕ\nh : n ≤ m\n⊢ n + 0 ≤ m" + synth.synth_text(rng, 30)
    t0 = time.time()
    emb = _ref_encode(model, texts, 1024, 4)
    print(f"g9: HF fp32 encode of {sum(lens)} tokens in {time.time() - t0:.1f}s")
    o_emb = t5_ref.encode_texts(cfg, sd, texts, 1024, 4)
    de = (emb - o_emb).abs().max().item()
    print(f"g9: oracle vs HF max|Δemb| = {de:.2e}")
    assert de < 5e-5
    model_bf = model.to(torch.bfloat16)
    emb_bf = _ref_encode(model_bf, texts, 1024, 4).float()
    print(f"g9: HF bf16 vs fp32 max|Δemb| = {(emb_bf - emb).abs().max().item():.2e}; "
          f"min cos = {torch.nn.functional.cosine_similarity(emb_bf, emb).min().item():.5f}")
    np.savez_compressed(os.path.join(OUT, "g9_byt5_base.npz" if scale == "sharp" else "g9h_byt5_base.npz"),
                        texts=np.array(texts, dtype=object), emb=emb.numpy(),
                        emb_hf_bf16=emb_bf.numpy().astype(np.float16), seed=np.int64(synth.SEED),
                        **({} if scale == "sharp" else {"weight_scale": np.array(scale)}))
    print(f"g9[{scale}] ok")


# ----------------------------------------------------------------------------------------------
def g10_train_forward():
    """The training forward: the reference's own ``collate`` (is_train=True: tokenisation + label
    matrix, datamodule.py:130-198) and ``forward`` (contrastive MSE, model.py:116-140) on a hand-built batch -
    6 examples, 3 negatives each, one example's negative being another's positive, duplicates in all_pos."""
    import importlib
    from types import SimpleNamespace
    from oracle import train_ref

    dmod = importlib.import_module("retrieval.datamodule")
    cfg = synth.t5_config("byt5-small")
    cfg["num_layers"] = 4  # the loss logic is what this fixture pins; 4 layers keep the fp32 CPU run short
    sd = synth.synth_state_dict(cfg, seed=10)
    model = H.offline_retriever(rm, hf_cfg(cfg), sd, max_seq_len=512)
    files = synth.synth_corpus_records(20, 300, seed=101, code_bytes=(30, 160))
    td = tempfile.mkdtemp()
    path = os.path.join(td, "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    corpus = common.Corpus(path)
    prem = corpus.all_premises
    where = {id(p): i for i, p in enumerate(prem)}
    rng = np.random.default_rng(102)
    n, nneg = 6, 3
    pick = rng.choice(len(prem), size=n * (2 + nneg), replace=False)
    examples = []
    for j in range(n):
        pos = prem[int(pick[j])]
        extra = prem[int(pick[n + j])]
        negs = [prem[int(pick[2 * n + j * nneg + i])] for i in range(nneg)]
        state = synth.synth_state(rng, int(rng.integers(40, 220)))
        examples.append({"context": common.Context(pos.path, f"thm{j}", H.Pos(500, 0), state), "pos_premise": pos,
                         "all_pos_premises": [pos, extra], "neg_premises": negs})
    examples[1]["neg_premises"][0] = examples[0]["pos_premise"]        # another example's positive as a negative
    examples[2]["all_pos_premises"].append(examples[3]["pos_premise"])  # ... and as a further positive
    examples[4]["all_pos_premises"].append(examples[5]["neg_premises"][2])
    fake_self = SimpleNamespace(tokenizer=model.tokenizer, max_seq_len=512, num_negatives=nneg, is_train=True)
    batch = dmod.RetrievalDataset.collate(fake_self, examples)
    label = batch["label"]
    with torch.no_grad():
        loss = model(batch["context_ids"], batch["context_mask"], batch["pos_premise_ids"], batch["pos_premise_mask"],
                     batch["neg_premises_ids"], batch["neg_premises_mask"], label)
        ctx = model._encode(batch["context_ids"], batch["context_mask"])
        allp = torch.cat([model._encode(batch["pos_premise_ids"], batch["pos_premise_mask"])] +
                         [model._encode(i, m) for i, m in zip(batch["neg_premises_ids"], batch["neg_premises_mask"])])
        sim = ctx @ allp.T
    # the oracle restatement
    o_label = train_ref.label_matrix([where[id(e["pos_premise"])] for e in examples],
                                     [[where[id(p)] for p in e["neg_premises"]] for e in examples],
                                     [[where[id(p)] for p in e["all_pos_premises"]] for e in examples])
    assert np.array_equal(o_label, label.numpy()), "oracle label-matrix drift"
    ctx_texts = [e["context"].serialize() for e in examples]
    pos_texts = [e["pos_premise"].serialize() for e in examples]
    neg_texts = [[e["neg_premises"][i].serialize() for e in examples] for i in range(nneg)]
    o_loss, o_sim = train_ref.forward_loss(cfg, sd, ctx_texts, pos_texts, neg_texts, o_label, 512)
    print(f"g10: reference loss {float(loss):.6f}; oracle {o_loss:.6f}; max|Δsim| {np.abs(o_sim - sim.numpy()).max():.2e}; "
          f"label ones {int(label.sum())} of {label.numel()}")
    assert abs(o_loss - float(loss)) < 1e-6 and np.abs(o_sim - sim.numpy()).max() < 2e-5
    np.savez_compressed(
        os.path.join(OUT, "g10_train_forward.npz"), context_texts=np.array(ctx_texts, dtype=object),
        pos_texts=np.array(pos_texts, dtype=object), neg_texts=np.array(neg_texts, dtype=object),
        pos_idx=np.array([where[id(e["pos_premise"])] for e in examples]),
        neg_idx=np.array([[where[id(p)] for p in e["neg_premises"]] for e in examples]),
        all_pos_idx=np.array([[where[id(p)] for p in e["all_pos_premises"]] + [-1] * (4 - len(e["all_pos_premises"]))
                              for e in examples]),
        label=label.numpy(), loss=np.float64(float(loss)), similarity=sim.numpy(), num_layers=np.int64(4),
        weight_seed=np.int64(10), corpus_seed=np.int64(101), max_seq_len=np.int64(512))
    print("g10 ok")


def g11_train_backward():
    """The training step: ``loss.backward()`` through the reference's ``forward`` (model.py:116-140) on
    its own collated batch, then two optimizer steps as ``get_optimizers`` configures them outside DeepSpeed
    (common.py:395-398: torch.optim.AdamW(lr) + get_constant_schedule_with_warmup).  Tiny geometry (2 layers, d_model
    128) so that every gradient fits in a fixture; dropout off (eval mode): T5's dropout is stochastic."""
    import importlib
    from types import SimpleNamespace

    from transformers import get_constant_schedule_with_warmup

    from oracle import train_ref

    dmod = importlib.import_module("retrieval.datamodule")
    cfg = synth.t5_config("tiny")
    sd = synth.synth_state_dict(cfg, seed=11)
    model = H.offline_retriever(rm, hf_cfg(cfg), sd, max_seq_len=256)
    model.eval()
    files = synth.synth_corpus_records(12, 120, seed=111, code_bytes=(20, 90))
    td = tempfile.mkdtemp()
    path = os.path.join(td, "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    corpus = common.Corpus(path)
    prem = corpus.all_premises
    where = {id(p): i for i, p in enumerate(prem)}
    rng = np.random.default_rng(112)
    n, nneg = 4, 2
    pick = rng.choice(len(prem), size=n * (2 + nneg), replace=False)
    examples = []
    for j in range(n):
        pos = prem[int(pick[j])]
        extra = prem[int(pick[n + j])]
        negs = [prem[int(pick[2 * n + j * nneg + i])] for i in range(nneg)]
        state = synth.synth_state(rng, int(rng.integers(30, 120)))
        examples.append({"context": common.Context(pos.path, f"thm{j}", H.Pos(500, 0), state), "pos_premise": pos,
                         "all_pos_premises": [pos, extra], "neg_premises": negs})
    examples[1]["neg_premises"][0] = examples[0]["pos_premise"]
    fake_self = SimpleNamespace(tokenizer=model.tokenizer, max_seq_len=256, num_negatives=nneg, is_train=True)
    batch = dmod.RetrievalDataset.collate(fake_self, examples)
    label = batch["label"]
    lr, warmup = 1e-3, 1
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=lr)
    sched = get_constant_schedule_with_warmup(opt, warmup)
    names = {id(p): k for k, p in model.named_parameters()}

    def strip(k):  # PremiseRetriever.encoder.<HF name>
        assert k.startswith("encoder."), k
        return k[len("encoder."):]

    def step():
        opt.zero_grad()
        loss = model(batch["context_ids"], batch["context_mask"], batch["pos_premise_ids"], batch["pos_premise_mask"],
                     batch["neg_premises_ids"], batch["neg_premises_mask"], label)
        loss.backward()
        grads = {strip(names[id(p)]): p.grad.detach().clone().numpy() for p in params}
        opt.step()
        sched.step()
        return float(loss.detach()), grads

    loss0, grads0 = step()  # learning rate 0 (warm-up): parameters unchanged, moments updated
    loss1, grads1 = step()  # learning rate lr
    loss2, _ = step()
    after = {strip(k): p.detach().clone().numpy() for k, p in model.named_parameters()}
    # ---- the oracle restatement against the reference, here and now
    ctx_texts = [e["context"].serialize() for e in examples]
    pos_texts = [e["pos_premise"].serialize() for e in examples]
    neg_texts = [[e["neg_premises"][i].serialize() for e in examples] for i in range(nneg)]
    o_label = train_ref.label_matrix([where[id(e["pos_premise"])] for e in examples],
                                     [[where[id(p)] for p in e["neg_premises"]] for e in examples],
                                     [[where[id(p)] for p in e["all_pos_premises"]] for e in examples])
    assert np.array_equal(o_label, label.numpy())
    o_loss, o_grads = train_ref.forward_backward(cfg, sd, ctx_texts, pos_texts, neg_texts, o_label, 256)
    assert set(o_grads) == set(grads0), (sorted(set(o_grads) ^ set(grads0)))
    worst = max(np.abs(o_grads[k] - grads0[k]).max() / (np.abs(grads0[k]).max() + 1e-12) for k in grads0)
    print(f"g11: loss {loss0:.6f} (oracle {o_loss:.6f}); worst relative gradient difference oracle vs reference {worst:.2e}; "
          f"losses over the three steps {loss0:.6f} {loss1:.6f} {loss2:.6f}")
    assert abs(o_loss - loss0) < 1e-6 and worst < 2e-4
    # AdamW restatement: three steps from the reference's gradients of steps 0, 1 (step 2's are not stored: the
    # parameters after the SECOND effective update are what the fixture pins) - replay steps 0..1 and compare with the
    # reference's parameters before its third update... simpler and sufficient: replay all three with recomputed grads
    P = {k: v.numpy().astype(np.float32).copy() for k, v in sd.items() if k != train_ref.TIED}
    M = {k: np.zeros_like(v, dtype=np.float64) for k, v in P.items()}
    V = {k: np.zeros_like(v, dtype=np.float64) for k, v in P.items()}
    for t in range(3):
        _, g = train_ref.forward_backward(cfg, {k: torch.from_numpy(v) for k, v in P.items()}, ctx_texts, pos_texts,
                                          neg_texts, o_label, 256)
        for k in P:
            P[k], M[k], V[k] = train_ref.adamw_step(P[k], g[k], M[k], V[k], t + 1, lr * train_ref.warmup_factor(t, warmup))
    worst_p = max(np.abs(P[k] - after[k]).max() for k in P)
    print(f"g11: max |parameter difference| after three optimizer steps, oracle vs reference: {worst_p:.2e}")
    assert worst_p < 5e-6
    out = {"loss": np.float64(loss0), "losses": np.array([loss0, loss1, loss2]), "label": label.numpy(),
           "context_texts": np.array(ctx_texts, dtype=object), "pos_texts": np.array(pos_texts, dtype=object),
           "neg_texts": np.array(neg_texts, dtype=object), "weight_seed": np.int64(11), "max_seq_len": np.int64(256),
           "lr": np.float64(lr), "warmup_steps": np.int64(warmup)}
    for k, v in grads0.items():
        out["grad/" + k] = v.astype(np.float32)
    for k, v in after.items():
        out["after3/" + k] = v.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "g11_train_backward.npz"), **out)
    print("g11 ok")


def g13_train_examples():
    """Training-side data loading: the reference's RetrievalDataset with is_train=True - one example per (tactic,
    positive premise) (datamodule.py:60-75) and the candidate pools of the negative sampling in ``__getitem__``
    (:95-128).  The pools' ORDER is not reproducible in the reference itself (networkx builds the closure's successor
    lists from Python sets of path strings), so the fixture pins what is: the pools as sets, and how many negatives
    come from each (``random.sample`` is intercepted to record its populations)."""
    import importlib
    import random as _random

    dmod = importlib.import_module("retrieval.datamodule")
    files = synth.synth_corpus_records(30, 500, seed=131, max_imports=5)
    td = tempfile.mkdtemp()
    cpath = os.path.join(td, "corpus.jsonl")
    synth.write_corpus_jsonl(cpath, files)
    split = synth.synth_split(files, 40, seed=132, min_file=8)
    spath = os.path.join(td, "train.json")
    json.dump(split, open(spath, "w"))
    corpus = common.Corpus(cpath)
    tok = ByT5Tokenizer()
    num_neg, num_in_file = 3, 1
    ds = dmod.RetrievalDataset([spath], corpus, num_neg, num_in_file, 256, tok, is_train=True)
    where = {id(p): i for i, p in enumerate(corpus.all_premises)}
    calls = []
    real_sample = _random.sample

    def spy(population, k):
        calls.append(([where[id(p)] for p in population], k))
        return real_sample(population, k)

    examples = []
    dmod.random.sample = spy
    try:
        _random.seed(133)
        for i, ex in enumerate(ds.data):
            calls.clear()
            raises = False
            try:
                got = ds[i]
                assert len(got["neg_premises"]) == num_neg
            except ValueError:  # random.sample: pool smaller than the request (the reference does not guard it)
                raises = True
            assert len(calls) == 2
            (in_file, k_in), (outside, k_out) = calls
            examples.append({"raises": raises,"file_path": ex["file_path"], "full_name": ex["full_name"], "start": ex["start"],
                             "tactic_idx": ex["tactic_idx"], "pos_premise": where[id(ex["pos_premise"])],
                             "all_pos_premises": sorted(where[id(p)] for p in ex["all_pos_premises"]),
                             "in_file_pool": sorted(in_file), "in_file_multiset": len(in_file) != len(set(in_file)),
                             "outside_pool": sorted(outside), "k_in": k_in, "k_out": k_out})
    finally:
        dmod.random.sample = real_sample
    # the reference lists a tactic's positives in set-iteration order: the fixture is keyed, not ordered
    examples.sort(key=lambda e: (e["full_name"], e["tactic_idx"], e["pos_premise"]))
    json.dump({"corpus_seed": 131, "n_files": 30, "n_premises": 500, "max_imports": 5, "split_seed": 132, "n_theorems": 40,
               "min_file": 8, "num_negatives": num_neg, "num_in_file_negatives": num_in_file, "examples": examples},
              open(os.path.join(OUT, "g13_train_examples.json"), "w"))
    n_in = sum(1 for e in examples if e["k_in"])
    print(f"g13 ok: {len(examples)} training examples, {n_in} with an in-file negative, "
          f"{sum(e['raises'] for e in examples)} whose pools are too small (ValueError); "
          f"pool sizes in-file up to {max(len(e['in_file_pool']) for e in examples)}, outside up to "
          f"{max(len(e['outside_pool']) for e in examples)}")


def g14_reference_indexed_corpus():
    """The reference's own index file: ``IndexedCorpus(corpus, embeddings)`` pickled exactly as retrieval/index.py:37-40
    writes it (the reference's common.Corpus with its networkx graph, common.File / Premise, lean_dojo.Pos inside), for a
    small corpus and the tiny encoder, plus what the reference's ``retrieve`` returns from it for a few states.  The
    product reads this file without any of those modules (common.load_indexed_corpus_pickle).
    Regenerating reproduces the file except for 22 bytes: the storage key torch's legacy tensor pickling writes twice
    is a heap address of the generating process (two 14-digit decimal strings); everything else is byte-identical."""
    import pickle

    cfg = synth.t5_config("tiny")
    sd = synth.synth_state_dict(cfg, seed=14)
    model = H.offline_retriever(rm, hf_cfg(cfg), sd, max_seq_len=512)
    files = synth.synth_corpus_records(12, 120, seed=141, max_imports=4, code_bytes=(20, 90))
    td = tempfile.mkdtemp()
    path = os.path.join(td, "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    model.load_corpus(path)
    model.reindex_corpus(batch_size=32)
    H.Pos.__module__ = "lean_dojo"  # where the stand-in is importable from, as the real class is from its package
    out_pickle = os.path.join(OUT, "g14_reference_indexed_corpus.pickle")
    with open(out_pickle, "wb") as fh:  # index.py:37-40
        pickle.dump(common.IndexedCorpus(model.corpus, model.corpus_embeddings.to(torch.float32).cpu()), fh)
    where = {id(p): i for i, p in enumerate(model.corpus.all_premises)}
    rng = np.random.default_rng(142)
    queries = []
    for j in range(6):
        while True:
            f = int(rng.integers(6, 12))
            pos = (int(rng.integers(1, 300)), int(rng.integers(0, 40)))
            if len(model.corpus.get_accessible_premises(files[f]["path"], H.Pos(*pos))) >= 8:
                break
        state = synth.synth_state(rng, int(rng.integers(40, 160)))
        prem, sc = model.retrieve(state, files[f]["path"], f"thm{j}", H.Pos(*pos), 5)
        queries.append({"path": files[f]["path"], "pos": list(pos), "state": state, "ids": [where[id(p)] for p in prem],
                        "scores": sc})
    json.dump({"corpus_seed": 141, "n_files": 12, "n_premises": 120, "max_imports": 4, "code_bytes": [20, 90],
               "weight_seed": 14, "N": len(model.corpus), "queries": queries, "k": 5},
              open(os.path.join(OUT, "g14_reference_indexed_corpus.json"), "w"), ensure_ascii=False)
    print(f"g14 ok: {len(model.corpus)} premises, pickle {os.path.getsize(out_pickle)} bytes")


def g17_reference_loads_our_pickle():
    """The other direction of G14: an index file written by THIS package under the reference's class names
    (reprover_amd.common.save_reference_pickle, `retrieval/index.py --reference-pickle`) is loaded by the imported
    reference with plain pickle (retrieval/model.py:81-85, as prover/tactic_generator.py:273-276 does) and its
    ``retrieve`` must return, from that file, exactly what it returns from the index it built itself.  Asserted here
    (authoring container); the JSON records the queries and the reference's answers from OUR file, and the CPU suite
    re-reads a freshly written file with the package's own loader against them."""
    import pickle

    from reprover_amd import common as our_common

    cfg = synth.t5_config("tiny")
    sd = synth.synth_state_dict(cfg, seed=14)
    model = H.offline_retriever(rm, hf_cfg(cfg), sd, max_seq_len=512)
    files = synth.synth_corpus_records(12, 120, seed=171, max_imports=4, code_bytes=(20, 90))
    td = tempfile.mkdtemp()
    path = os.path.join(td, "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    model.load_corpus(path)
    model.reindex_corpus(batch_size=32)
    E = model.corpus_embeddings.to(torch.float32).cpu()
    ours = os.path.join(td, "ours_for_the_reference.pickle")
    our_common.save_reference_pickle(ours, our_common.Corpus(path), E)
    with open(ours, "rb") as fh:
        loaded = pickle.load(fh)  # plain pickle: resolves common.* to the REFERENCE's classes, Pos to the harness' lean_dojo
    assert type(loaded) is common.IndexedCorpus and type(loaded.corpus) is common.Corpus
    assert type(loaded.corpus.all_premises[0]) is common.Premise and type(loaded.corpus.all_premises[0].start) is H.Pos
    assert [(p.path, p.full_name, tuple(p.start), tuple(p.end), p.code) for p in loaded.corpus.all_premises] == \
        [(p.path, p.full_name, tuple(p.start), tuple(p.end), p.code) for p in model.corpus.all_premises]
    g_ref, g_our = model.corpus.transitive_dep_graph, loaded.corpus.transitive_dep_graph
    assert list(g_ref.nodes) == list(g_our.nodes) and set(g_ref.edges) == set(g_our.edges)
    model2 = H.offline_retriever(rm, hf_cfg(cfg), sd, max_seq_len=512)
    model2.load_corpus(ours)  # retrieval/model.py:81-85
    assert not model2.embeddings_staled and torch.equal(model2.corpus_embeddings, E)
    where = {(p.path, p.full_name, tuple(p.start)): i for i, p in enumerate(model.corpus.all_premises)}
    rng = np.random.default_rng(172)
    queries = []
    for j in range(8):
        while True:
            f = int(rng.integers(6, 12))
            pos = (int(rng.integers(1, 300)), int(rng.integers(0, 40)))
            if len(model.corpus.get_accessible_premises(files[f]["path"], H.Pos(*pos))) >= 8:
                break
        state = synth.synth_state(rng, int(rng.integers(40, 160)))
        prem, sc = model.retrieve(state, files[f]["path"], f"thm{j}", H.Pos(*pos), 5)
        prem2, sc2 = model2.retrieve(state, files[f]["path"], f"thm{j}", H.Pos(*pos), 5)
        ids = [where[(p.path, p.full_name, tuple(p.start))] for p in prem]
        assert ids == [where[(p.path, p.full_name, tuple(p.start))] for p in prem2] and sc == sc2, j
        assert len(model.corpus.get_accessible_premises(files[f]["path"], H.Pos(*pos))) == \
            len(model2.corpus.get_accessible_premises(files[f]["path"], H.Pos(*pos)))
        queries.append({"path": files[f]["path"], "pos": list(pos), "state": state, "ids": ids, "scores": sc})
    json.dump({"corpus_seed": 171, "n_files": 12, "n_premises": 120, "max_imports": 4, "code_bytes": [20, 90],
               "weight_seed": 14, "N": len(model.corpus), "queries": queries, "k": 5,
               "checked": "the imported reference unpickled a file written by reprover_amd.common.save_reference_pickle with "
                          "plain pickle.load, load_corpus()ed it and retrieve()d these answers - identical to its own index"},
              open(os.path.join(OUT, "g17_reference_loads_our_pickle.json"), "w"), ensure_ascii=False)
    print(f"g17 ok: the reference loaded our pickle ({os.path.getsize(ours)} bytes) and retrieved identically")


def g12_train_small_width():
    """The training step at ByT5-small WIDTH (d_model 1472, 6 heads, d_ff 3584; 2 layers): the reference's
    ``loss.backward()`` and three AdamW steps as in G11, on a batch whose sequences span several 128-token blocks.  36 M
    gradients do not fit a fixture: every tensor is pinned by its L2 norm and by its values at 2048 seeded positions
    (all of them for the small tensors), likewise the parameters after the three steps."""
    import importlib
    from types import SimpleNamespace

    from transformers import get_constant_schedule_with_warmup

    from oracle import train_ref

    dmod = importlib.import_module("retrieval.datamodule")
    cfg = synth.t5_config("byt5-small")
    cfg["num_layers"] = 2
    sd = synth.synth_state_dict(cfg, seed=12)
    model = H.offline_retriever(rm, hf_cfg(cfg), sd, max_seq_len=512)
    model.eval()
    files = synth.synth_corpus_records(12, 120, seed=121, code_bytes=(30, 200))
    td = tempfile.mkdtemp()
    path = os.path.join(td, "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    corpus = common.Corpus(path)
    prem = corpus.all_premises
    where = {id(p): i for i, p in enumerate(prem)}
    rng = np.random.default_rng(122)
    n, nneg = 4, 2
    pick = rng.choice(len(prem), size=n * (2 + nneg), replace=False)
    examples = []
    for j in range(n):
        pos = prem[int(pick[j])]
        extra = prem[int(pick[n + j])]
        negs = [prem[int(pick[2 * n + j * nneg + i])] for i in range(nneg)]
        state = synth.synth_state(rng, int(rng.integers(60, 420)))
        examples.append({"context": common.Context(pos.path, f"thm{j}", H.Pos(500, 0), state), "pos_premise": pos,
                         "all_pos_premises": [pos, extra], "neg_premises": negs})
    examples[1]["neg_premises"][0] = examples[0]["pos_premise"]
    fake_self = SimpleNamespace(tokenizer=model.tokenizer, max_seq_len=512, num_negatives=nneg, is_train=True)
    batch = dmod.RetrievalDataset.collate(fake_self, examples)
    label = batch["label"]
    lr, warmup = 1e-3, 1
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=lr)
    sched = get_constant_schedule_with_warmup(opt, warmup)
    names = {id(p): k for k, p in model.named_parameters()}

    def strip(k):
        assert k.startswith("encoder."), k
        return k[len("encoder."):]

    def step(keep):
        opt.zero_grad()
        loss = model(batch["context_ids"], batch["context_mask"], batch["pos_premise_ids"], batch["pos_premise_mask"],
                     batch["neg_premises_ids"], batch["neg_premises_mask"], label)
        loss.backward()
        grads = {strip(names[id(p)]): p.grad.detach().clone().numpy() for p in params} if keep else None
        opt.step()
        sched.step()
        return float(loss.detach()), grads

    loss0, grads0 = step(True)
    loss1, _ = step(False)
    loss2, _ = step(False)
    after = {strip(k): p.detach().numpy() for k, p in model.named_parameters()}
    ctx_texts = [e["context"].serialize() for e in examples]
    pos_texts = [e["pos_premise"].serialize() for e in examples]
    neg_texts = [[e["neg_premises"][i].serialize() for e in examples] for i in range(nneg)]
    o_label = train_ref.label_matrix([where[id(e["pos_premise"])] for e in examples],
                                     [[where[id(p)] for p in e["neg_premises"]] for e in examples],
                                     [[where[id(p)] for p in e["all_pos_premises"]] for e in examples])
    assert np.array_equal(o_label, label.numpy())
    o_loss, o_grads = train_ref.forward_backward(cfg, sd, ctx_texts, pos_texts, neg_texts, o_label, 512)
    worst = max(np.abs(o_grads[k] - grads0[k]).max() / (np.abs(grads0[k]).max() + 1e-12) for k in grads0)
    lens = [int(m.sum()) for m in batch["context_mask"]] + [int(m.sum()) for m in batch["pos_premise_mask"]]
    print(f"g12: loss {loss0:.6f} (oracle {o_loss:.6f}); worst relative gradient difference oracle vs reference {worst:.2e}; "
          f"losses {loss0:.6f} {loss1:.6f} {loss2:.6f}; context/positive lengths {lens}")
    assert abs(o_loss - loss0) < 1e-6 and worst < 5e-4
    out = {"loss": np.float64(loss0), "losses": np.array([loss0, loss1, loss2]), "label": label.numpy(),
           "context_texts": np.array(ctx_texts, dtype=object), "pos_texts": np.array(pos_texts, dtype=object),
           "neg_texts": np.array(neg_texts, dtype=object), "weight_seed": np.int64(12), "max_seq_len": np.int64(512),
           "lr": np.float64(lr), "warmup_steps": np.int64(warmup), "num_layers": np.int64(2)}
    srng = np.random.default_rng(123)
    for k in sorted(grads0):
        g = grads0[k].reshape(-1)
        idx = np.sort(srng.choice(g.size, size=min(g.size, 2048), replace=False)).astype(np.int64)
        out["idx/" + k] = idx
        out["grad/" + k] = g[idx].astype(np.float32)
        out["gradnorm/" + k] = np.float64(np.linalg.norm(g.astype(np.float64)))
        out["after3/" + k] = after[k].reshape(-1)[idx].astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "g12_train_small_width.npz"), **out)
    print("g12 ok")

# ----------------------------------------------------------------------------------------------
def g15_augmented_state():
    """The reference's ``format_augmented_state`` (common.py:357-378): the consumer of ``retrieve`` on the prover side.
    Cases: no budget, budgets that cut the list in the middle (a later, shorter premise still fits: ``continue``, not
    ``break``), a budget smaller than the state, multi-byte text, and seeded ``p_drop`` (one ``random.random()`` per
    premise, in list order)."""
    import random

    P = H.Pos
    codes = [
        ("Nat.add_comm", "theorem Nat.add_comm (n m : ℕ) : n + m = m + n := sorry"),
        ("Foo.bar.baz", "lemma baz : 1 = 1\nlemma bar.baz' : 2 = 2"),
        ("x", "def x := x + x"),
        ("List.map", "def map (f : α → β) : List α → List β\n  | [] => []\n  | a::as => f a :: map f as"),
        ("Set.mem_def", "theorem mem_def {a : α} {s : Set α} : a ∈ s ↔ s a"),
        ("y", "def y := 0"),
        ("Real.sqrt", "noncomputable def Real.sqrt (x : ℝ) : ℝ := NNReal.sqrt (Real.toNNReal x)"),
    ]
    ref_p = [common.Premise("A.lean", n, P(1, 0), P(2, 0), c) for n, c in codes]
    states = ["n m : ℕ\n⊢ n + m = m + n", "⊢ True", "α : Type\ns : Set α\na : α\nh : a ∈ s\n⊢ s a"]
    cases = []
    for s in states:
        sb = len(s.encode("utf-8"))
        sizes = [len((p.serialize() + "\n\n").encode("utf-8")) for p in ref_p]
        budgets = [None, 0, sb - 1, sb, sb + sizes[0] - 1, sb + sizes[0], sb + sizes[0] + sizes[2],
                   sb + sum(sizes) - 1, sb + sum(sizes), 2300]
        for max_len in budgets:
            for p_drop, seed in ((0.0, 0), (0.5, 3407), (0.9, 11)):
                random.seed(seed)
                want = common.format_augmented_state(s, ref_p, max_len, p_drop)
                random.seed(seed)
                mine = common_ref.format_augmented_state(s, [p.serialize() for p in ref_p], max_len, p_drop)
                assert mine == want, (s, max_len, p_drop)
                cases.append({"state": s, "max_len": max_len, "p_drop": p_drop, "seed": seed, "out": want})
    json.dump({"premises": [{"path": "A.lean", "full_name": n, "code": c} for n, c in codes], "cases": cases},
              open(os.path.join(OUT, "g15_augmented_state.json"), "w"), ensure_ascii=False)
    print("g15 ok:", len(cases), "cases")



if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15",
                             "g5h", "g7h", "g9h", "g17"]
    torch.manual_seed(0)
    for name in which:
        {"g1": g1_tokenizer, "g2": g2_serialize, "g3": g3_buckets, "g4": g4_tiny, "g5": g5_small,
         "g6": g6_nearest, "g7": g7_predict, "g8": g8_eval_data, "g9": g9_base_full_depth, "g10": g10_train_forward, "g11": g11_train_backward,
         "g12": g12_train_small_width, "g13": g13_train_examples,
         "g14": g14_reference_indexed_corpus, "g15": g15_augmented_state,
         "g17": g17_reference_loads_our_pickle,
         "g5h": lambda: g5_small("hf"), "g7h": lambda: g7_predict("hf"), "g9h": lambda: g9_base_full_depth("hf")}[name]()
