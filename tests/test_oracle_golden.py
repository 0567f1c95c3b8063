"""The oracle (oracle/) against the golden vectors produced by the reference + HuggingFace
(tests/golden/make_golden.py).  CPU only.  If these fail the oracle has drifted from the reference and
no GPU parity claim means anything."""
import json
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import common_ref, t5_ref
from reprover_amd import synth


def test_g1_tokenizer(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "g1_tokenizer.json")))
    for case in g["cases"]:
        ids, mask = t5_ref.byt5_batch(g["texts"], case["max_length"])
        assert ids.shape[1] == case["padded_len"]
        for i, row in enumerate(case["ids"]):
            n = int(mask[i].sum())
            assert ids[i, :n].tolist() == row
            assert not ids[i, n:].any() and not mask[i, n:].any()


def test_g2_serialize_and_file_filters(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "g2_serialize.json")))
    P = common_ref.Pos
    for c in g["cases"]:
        p = common_ref.PremiseRef(c["path"], c["full_name"], P(1, 0), P(2, 0), c["code"])
        assert p.serialize() == c["serialized"]
    files = synth.synth_corpus_records(g["corpus_files"], g["corpus_premises"], seed=g["corpus_seed"])
    for fd, kept in zip(files, g["kept"]):
        prem = common_ref.premises_of_file(fd)
        assert [p.full_name for p in prem] == kept["names"]
        assert [p.serialize() for p in prem] == kept["serialized"]


def test_g3_buckets(golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_buckets.npz"))
    assert np.array_equal(t5_ref.relative_position_bucket(g["rel"], 32, 128), g["bucket"])


def test_g4_tiny_encoder(golden_dir, tiny_weights):
    cfg, sd = tiny_weights
    g = np.load(os.path.join(golden_dir, "g4_tiny.npz"), allow_pickle=True)
    ids, mask = g["input_ids"].astype(np.int64), g["attention_mask"].astype(np.int64)
    o_ids, o_mask = t5_ref.byt5_batch(list(g["texts"]), 512)
    assert np.array_equal(ids, o_ids) and np.array_equal(mask, o_mask)
    hidden = t5_ref.encoder_forward(cfg, sd, ids, mask)
    last = hidden[torch.arange(len(ids)), torch.from_numpy(mask.sum(1) - 1)]
    assert (last - torch.from_numpy(g["hidden_last_rows"])).abs().max() < 5e-5
    assert (hidden[:, 0] - torch.from_numpy(g["hidden_first_rows"])).abs().max() < 5e-5
    emb = t5_ref.encode(cfg, sd, ids, mask)
    assert (emb - torch.from_numpy(g["emb"])).abs().max() < 1e-5  # stated CPU-restatement tolerance


def test_g5_byt5_small(golden_dir, small_weights):
    cfg, sd = small_weights
    g = np.load(os.path.join(golden_dir, "g5_byt5_small.npz"), allow_pickle=True)
    texts = list(g["texts"])
    sub = [0, 3, 6, 9, 12]  # a few rows keep the CPU suite short; all rows are checked on the GPU
    emb = t5_ref.encode_texts(cfg, sd, [texts[i] for i in sub], 2048, 1)
    assert (emb - torch.from_numpy(g["emb"][sub])).abs().max() < 1e-5


def test_g5h_byt5_small_hf_init_scales(golden_dir, small_weights_hf):
    """The oracle against fixture G5h (the reference + HuggingFace fp32 on weights at HF's init scales)."""
    cfg, sd = small_weights_hf
    g = np.load(os.path.join(golden_dir, "g5h_byt5_small.npz"), allow_pickle=True)
    assert str(g["weight_scale"]) == "hf"
    texts = list(g["texts"])
    sub = [1, 4, 7]
    emb = t5_ref.encode_texts(cfg, sd, [texts[i] for i in sub], 2048, 1)
    assert (emb - torch.from_numpy(g["emb"][sub])).abs().max() < 1e-5
    # on this family HuggingFace's own bf16 mode - the reference's GPU numerics - meets the written contract
    hf = torch.from_numpy(g["emb_hf_bf16"].astype(np.float32))
    gold = torch.from_numpy(g["emb"])
    assert torch.nn.functional.cosine_similarity(hf, gold).min() >= 0.999
    assert ((hf @ hf.T) - (gold @ gold.T)).abs().max() <= 1e-2


def _g6_setup(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "g6_nearest.json")))
    z = np.load(os.path.join(golden_dir, "g6_nearest.npz"))
    files = synth.synth_corpus_records(g["n_files"], g["n_premises"], seed=g["corpus_seed"])
    td = tempfile.mkdtemp()
    path = os.path.join(td, "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    return g, z, path


def test_g6_nearest_premises(golden_dir):
    g, z, path = _g6_setup(golden_dir)
    corpus = common_ref.CorpusRef(path)
    assert len(corpus) == g["N"]
    E, Q = z["E"], z["Q"]
    acc = np.unpackbits(z["acc"], axis=1)[:, : g["N"]].astype(bool)
    ctxs = [
        common_ref.ContextRef(q["path"], f"thm{j}", common_ref.Pos(*q["pos"]), f"x{j} ⊢ y")
        for j, q in enumerate(g["queries"])
    ]
    for j, c in enumerate(ctxs):
        keys = corpus.accessible_keys(c.path, c.theorem_pos)
        row = np.array([(p.path, p.full_name) in keys for p in corpus.all_premises])
        assert np.array_equal(row, acc[j])
    for k, res in g["results"].items():
        qs = res["queries"]
        idx, sc = corpus.get_nearest_premises(E, [ctxs[j] for j in qs], Q[qs], int(k))
        assert idx == res["ids"]
        assert np.allclose(np.array(sc), np.array(res["scores"]), atol=1e-6)
        # array form used to check kernels
        ids2, sc2 = common_ref.masked_topk(Q[qs] @ E.T, acc[qs], int(k))
        assert ids2.tolist() == res["ids"]
    bad = g["value_error_query"]
    with pytest.raises(ValueError):
        corpus.get_nearest_premises(E, [ctxs[bad]], Q[[bad]], 100)


def test_format_augmented_state():
    s = "a ⊢ b"
    out = common_ref.format_augmented_state(s, ["p1", "p22", "p333"], max_len=len(s.encode()) + 9)
    # p1\n\n = 4 bytes, p22\n\n = 5 bytes fit (9); p333 does not; later premises go in front
    assert out == "p22\n\np1\n\n" + s


def test_g15_augmented_state(golden_dir):
    """oracle/common_ref.format_augmented_state against the reference's own outputs (G15)."""
    import json
    import random

    g = json.load(open(os.path.join(golden_dir, "g15_augmented_state.json")))
    texts = [common_ref.PremiseRef(p["path"], p["full_name"], common_ref.Pos(1, 0), common_ref.Pos(2, 0), p["code"]).serialize()
             for p in g["premises"]]
    for c in g["cases"]:
        random.seed(c["seed"])
        assert common_ref.format_augmented_state(c["state"], texts, c["max_len"], c["p_drop"]) == c["out"], c


def test_g10_train_forward(golden_dir):
    """oracle/train_ref.py (label matrix + contrastive-MSE forward) against the reference's own collate + forward."""
    from oracle import train_ref

    g = np.load(os.path.join(golden_dir, "g10_train_forward.npz"), allow_pickle=True)
    all_pos = [[int(x) for x in row if x >= 0] for row in g["all_pos_idx"]]
    label = train_ref.label_matrix(g["pos_idx"].tolist(), g["neg_idx"].tolist(), all_pos)
    assert np.array_equal(label, g["label"])
    cfg = synth.t5_config("byt5-small")
    cfg["num_layers"] = int(g["num_layers"])
    sd = synth.synth_state_dict(cfg, seed=int(g["weight_seed"]))
    loss, sim = train_ref.forward_loss(cfg, sd, list(g["context_texts"]), list(g["pos_texts"]),
                                       [list(r) for r in g["neg_texts"]], label, int(g["max_seq_len"]))
    assert abs(loss - float(g["loss"])) < 1e-6
    assert np.abs(sim - g["similarity"]).max() < 2e-5


def test_g11_train_backward_and_adamw(golden_dir):
    """oracle/train_ref.py (autograd over the forward restatement; AdamW + constant-with-warmup schedule) against the
    reference's own ``loss.backward()`` and three optimizer steps (common.py:381-405 outside DeepSpeed)."""
    import torch

    from oracle import train_ref

    g = np.load(os.path.join(golden_dir, "g11_train_backward.npz"), allow_pickle=True)
    cfg = synth.t5_config("tiny")
    sd = synth.synth_state_dict(cfg, seed=int(g["weight_seed"]))
    texts = (list(g["context_texts"]), list(g["pos_texts"]), [list(r) for r in g["neg_texts"]])
    L = int(g["max_seq_len"])
    loss, grads = train_ref.forward_backward(cfg, sd, *texts, g["label"], L)
    assert abs(loss - float(g["loss"])) < 1e-6
    gold = {k[len("grad/"):]: g[k] for k in g.files if k.startswith("grad/")}
    assert set(gold) == set(grads) and train_ref.TIED not in gold  # the tied embedding is one parameter
    for k, want in gold.items():
        assert np.abs(grads[k] - want).max() <= 1e-5 * np.abs(want).max() + 1e-9, k
    assert np.abs(gold["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]).max() > 0  # bias table trained
    # three optimizer steps on the same batch (step 0 at learning rate 0: warm-up), parameters after the third
    lr, warmup = float(g["lr"]), int(g["warmup_steps"])
    assert [train_ref.warmup_factor(t, warmup) for t in range(3)] == [0.0, 1.0, 1.0]
    P = {k: v.numpy().astype(np.float32).copy() for k, v in sd.items() if k != train_ref.TIED}
    M = {k: np.zeros(v.shape) for k, v in P.items()}
    V = {k: np.zeros(v.shape) for k, v in P.items()}
    losses = []
    for t in range(3):
        l, gr = train_ref.forward_backward(cfg, {k: torch.from_numpy(v) for k, v in P.items()}, *texts, g["label"], L)
        losses.append(l)
        for k in P:
            P[k], M[k], V[k] = train_ref.adamw_step(P[k], gr[k], M[k], V[k], t + 1, lr * train_ref.warmup_factor(t, warmup))
    assert np.abs(np.array(losses) - g["losses"]).max() < 2e-6 and losses[2] < losses[0]
    for k in P:
        assert np.abs(P[k] - g["after3/" + k]).max() < 5e-6, k


def test_g12_train_small_width(golden_dir):
    """The oracle's backward at ByT5-small width against the reference's sampled gradients (one forward + backward of a
    2-layer encoder on the CPU: a few seconds)."""
    from oracle import train_ref

    g = np.load(os.path.join(golden_dir, "g12_train_small_width.npz"), allow_pickle=True)
    cfg = synth.t5_config("byt5-small")
    cfg["num_layers"] = int(g["num_layers"])
    sd = synth.synth_state_dict(cfg, seed=int(g["weight_seed"]))
    texts = (list(g["context_texts"]), list(g["pos_texts"]), [list(r) for r in g["neg_texts"]])
    loss, grads = train_ref.forward_backward(cfg, sd, *texts, g["label"], int(g["max_seq_len"]))
    assert abs(loss - float(g["loss"])) < 1e-6
    keys = [k[len("grad/"):] for k in g.files if k.startswith("grad/")]
    assert set(keys) == set(grads)
    for k in keys:
        got, want = grads[k].reshape(-1)[g["idx/" + k]], g["grad/" + k]
        assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max() + 1e-10, k
        assert abs(np.linalg.norm(grads[k].astype(np.float64)) - float(g["gradnorm/" + k])) <= 1e-5 * float(g["gradnorm/" + k])
