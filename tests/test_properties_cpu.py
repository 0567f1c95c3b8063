"""Property tests of the host logic against the oracle (CPU only; hypothesis draws the cases the fixtures do not hold):
tokenizer, Premise.serialize, accessibility masks of random import DAGs, the prover-side formatter, shard bounds."""
import os
import random
import tempfile

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import common_ref, t5_ref
from reprover_amd import synth
from reprover_amd.common import Context, Corpus, Pos, Premise, format_augmented_state
from reprover_amd.dist import shard_bounds
from reprover_amd.tokenizer import ByT5Tokenizer, encode_packed

SETTINGS = dict(max_examples=150, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])

# text with the multi-byte symbols Lean states carry, specials of the tokenizer, and ordinary ASCII
_alphabet = st.sampled_from(list("abcXYZ 01_.:()\n\t«»ℕ⊢→∀αβ🙂") + ["</s>", "<pad>", "<unk>", "<extra_id_3>"])
_text = st.lists(_alphabet, min_size=0, max_size=60).map("".join)


@settings(**SETTINGS)
@given(st.lists(_text, min_size=1, max_size=6), st.sampled_from([4, 16, 33, 1024]))
def test_tokenizer_equals_oracle(texts, max_length):
    """ids / mask of the padded call form (model.py:199-205) and the packed form the engine consumes, against the oracle's
    restatement of HF's ByT5 tokenizer (pinned by G1) - incl. truncation in the middle of a multi-byte character."""
    want_ids, want_mask = t5_ref.byt5_batch(texts, max_length)
    got = ByT5Tokenizer()(texts, padding="longest", max_length=max_length, truncation=True, return_tensors="pt")
    assert np.array_equal(got.input_ids.numpy(), want_ids) and np.array_equal(got.attention_mask.numpy(), want_mask)
    ids, cu = encode_packed(texts, max_length)
    lens = want_mask.sum(1)
    assert np.array_equal(np.diff(cu), lens) and cu[0] == 0
    assert np.array_equal(ids, np.concatenate([want_ids[b, : lens[b]] for b in range(len(texts))]))


_name_part = st.sampled_from(["Nat", "add_comm", "foo", "a", "b", "x'", "«q»", "succ_le", "List", "map", "a.b", "(", "*", "+"])
_code_piece = st.sampled_from(["theorem ", "lemma ", "def ", " : ", ":= ", "\n", "\t", "ℕ", "_root_.", " ", "«", "»", "(", ")"]) | _name_part


@settings(**SETTINGS)
@given(st.lists(_name_part, min_size=1, max_size=4).map(".".join), st.lists(_code_piece, min_size=1, max_size=25).map("".join))
def test_serialize_equals_oracle(full_name, code):
    """<a>..</a> mark-up (common.py:93-106): the reference's unescaped-regex semantics, names with metacharacters included
    (a pattern the reference's ``re`` would reject is rejected the same way)."""
    try:
        want = common_ref.PremiseRef("A.lean", full_name, common_ref.Pos(1, 0), common_ref.Pos(2, 0), code).serialize()
    except Exception as exc:  # noqa: BLE001 - re.error from the reference's own unescaped pattern
        with pytest.raises(type(exc)):
            Premise("A.lean", full_name, Pos(1, 0), Pos(2, 0), code).serialize()
        return
    assert Premise("A.lean", full_name, Pos(1, 0), Pos(2, 0), code).serialize() == want


@st.composite
def _corpora(draw):
    n_files = draw(st.integers(1, 7))
    recs = []
    for f in range(n_files):
        imports = sorted(draw(st.sets(st.integers(0, f - 1), max_size=3))) if f else []
        prem, line = [], 1
        for j in range(draw(st.integers(0, 5))):
            start = (line, draw(st.integers(0, 5)))
            end = (line + draw(st.integers(0, 3)), draw(st.integers(0, 40)))
            if end < start:
                end = start
            name = draw(st.sampled_from(["foo", "bar", "baz", f"n{f}_{j}"]))  # duplicates inside and across files
            prem.append({"full_name": name, "code": f"theorem {name} : True", "start": list(start), "end": list(end),
                         "kind": "theorem"})
            line += draw(st.integers(0, 2))  # overlapping / nested declarations: `end <= pos` is not monotone
        recs.append({"path": f"M/F{f}.lean", "imports": [f"M/F{i}.lean" for i in imports], "premises": prem})
    return recs


@settings(**SETTINGS)
@given(_corpora(), st.data())
def test_accessibility_equals_oracle_on_random_import_dags(recs, data):
    """Array form of accessibility (closure bit rows + own-file end keys, SURVEY.md §8 a6') against the oracle's restatement
    of common.py:268-289 (set semantics on (path, full_name)) for random DAGs, duplicate names and nested declarations."""
    path = os.path.join(tempfile.mkdtemp(), "c.jsonl")
    synth.write_corpus_jsonl(path, recs)
    corpus, ref = Corpus(path), common_ref.CorpusRef(path)
    assert [p.full_name for p in corpus.all_premises] == [p.full_name for p in ref.all_premises]
    if not len(corpus):
        return
    ctxs = []
    for _ in range(4):
        f = data.draw(st.integers(0, len(recs) - 1))
        pos = (data.draw(st.integers(1, 12)), data.draw(st.integers(0, 45)))
        keys = ref.accessible_keys(recs[f]["path"], common_ref.Pos(*pos))
        want = np.array([(p.path, p.full_name) in keys for p in ref.all_premises])
        assert np.array_equal(corpus.accessible_mask(recs[f]["path"], Pos(*pos)), want), (f, pos)
        ctxs.append((Context(recs[f]["path"], "t", Pos(*pos), "a ⊢ b"), want))
    # the batch operands the kernels take reproduce the same masks
    bits_t, own, qk = corpus.query_masks([c for c, _ in ctxs])
    j = np.arange(len(ctxs))
    imported = (bits_t[corpus.file_of][:, j >> 5] >> (j & 31).astype(np.uint32)) & 1
    own_ok = (corpus.file_of[:, None] == own[None, :]) & (corpus.end_key[:, None] <= qk[None, :])
    got = imported.astype(bool) | own_ok
    # (the index form differs from the set form only for duplicate names inside one file, where the set form wins)
    for q, (_, want) in enumerate(ctxs):
        assert np.array_equal(got[:, q] | want, want) and np.array_equal(got[:, q], want & got[:, q])
        names = {}
        for i, p in enumerate(ref.all_premises):
            names.setdefault((p.path, p.full_name), []).append(i)
        for idx in names.values():  # per (path, name) group the array form marks exactly what the set form marks
            assert got[idx, q].any() == want[idx].any()


@settings(**SETTINGS)
@given(_text, st.lists(st.tuples(_name_part, _text.filter(lambda s: s != "")), min_size=0, max_size=6),
       st.one_of(st.none(), st.integers(0, 400)), st.sampled_from([0.0, 0.3, 1.0]), st.integers(0, 10_000))
def test_format_augmented_state_equals_oracle(state, prems, max_len, p_drop, seed):
    ps = [Premise("A.lean", n, Pos(1, 0), Pos(2, 0), c) for n, c in prems]
    try:
        texts = [p.serialize() for p in ps]
    except Exception:  # noqa: BLE001 - an unescaped-regex name the reference itself rejects
        return
    random.seed(seed)
    want = common_ref.format_augmented_state(state, texts, max_len, p_drop)
    random.seed(seed)
    assert format_augmented_state(state, ps, max_len, p_drop) == want


@settings(**SETTINGS)
@given(st.lists(st.integers(1, 2049), min_size=0, max_size=300), st.integers(1, 9))
def test_shard_bounds_are_contiguous_covering_and_balanced(weights, world):
    b = shard_bounds(weights, world)
    assert len(b) == world + 1 and b[0] == 0 and b[-1] == len(weights) and (np.diff(b) >= 0).all()
    if weights:
        w = np.asarray(weights)
        loads = np.array([w[b[r]: b[r + 1]].sum() for r in range(world)])
        # every shard is within one premise of the ideal share (cuts fall on cumulative-weight targets)
        assert (np.abs(np.cumsum(loads)[:-1] - w.sum() * np.arange(1, world) / world) <= w.max()).all()
