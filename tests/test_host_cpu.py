"""Host-side logic of the product package, checked on the CPU against the golden vectors and
the oracle: tokenizer, serialisation, corpus arrays / accessibility masks, the C-ABI surface.
No kernel is launched here."""
import ctypes
import json
import os
import re
import tempfile

import numpy as np
import pytest
import torch

from oracle import common_ref
from reprover_amd import _lib, synth, tokenizer
from reprover_amd.common import Context, Corpus, File, Pos, Premise, format_augmented_state

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tokenizer_matches_hf_golden(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "g1_tokenizer.json")))
    tok = tokenizer.ByT5Tokenizer()
    for case in g["cases"]:
        enc = tok(g["texts"], padding="longest", max_length=case["max_length"], truncation=True, return_tensors="pt")
        assert enc.input_ids.dtype == torch.int64 and enc.input_ids.shape[1] == case["padded_len"]
        ids, cu = tok.packed(g["texts"], case["max_length"])
        assert ids.dtype == np.int32 and cu[0] == 0 and cu[-1] == len(ids)
        for i, row in enumerate(case["ids"]):
            n = int(enc.attention_mask[i].sum())
            assert enc.input_ids[i, :n].tolist() == row
            assert int(enc.input_ids[i, n:].abs().sum()) == 0
            assert ids[cu[i] : cu[i + 1]].tolist() == row


def test_premise_serialize_and_file_filters(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "g2_serialize.json")))
    for c in g["cases"]:
        assert Premise(c["path"], c["full_name"], Pos(1, 0), Pos(2, 0), c["code"]).serialize() == c["serialized"]
    files = synth.synth_corpus_records(g["corpus_files"], g["corpus_premises"], seed=g["corpus_seed"])
    for fd, kept in zip(files, g["kept"]):
        f = File.from_data(fd)
        assert [p.full_name for p in f.premises] == kept["names"]
        assert [p.serialize() for p in f.premises] == kept["serialized"]


def test_pos_and_context_invariants():
    assert Pos(3, 4) < Pos(3, 5) < Pos(4, 0) and list(Pos(7, 9)) == [7, 9]
    assert Pos(3, 4).key() < Pos(3, 5).key() < Pos(4, 0).key()
    with pytest.raises(AssertionError):
        Context("A.lean", "t", Pos(1, 1), "no turnstile")
    with pytest.raises(AssertionError):
        Premise("A.lean", "x", Pos(2, 0), Pos(1, 0), "code")
    a = Context("A.lean", "t", Pos(1, 1), "x ⊢ y")
    b = Context("A.lean", "t", Pos(9, 9), "x ⊢ y")
    assert a == b and hash(a) == hash(b)  # theorem_pos is compare=False (common.py:40)


@pytest.fixture(scope="module")
def g6(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "g6_nearest.json")))
    z = np.load(os.path.join(golden_dir, "g6_nearest.npz"))
    files = synth.synth_corpus_records(g["n_files"], g["n_premises"], seed=g["corpus_seed"])
    path = os.path.join(tempfile.mkdtemp(), "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    return g, z, path


def test_corpus_arrays_reproduce_reference_accessibility(g6):
    g, z, path = g6
    corpus = Corpus(path)
    ref = common_ref.CorpusRef(path)
    assert len(corpus) == g["N"] == len(ref)
    assert [p.full_name for p in corpus.all_premises] == [p.full_name for p in ref.all_premises]
    acc = np.unpackbits(z["acc"], axis=1)[:, : g["N"]].astype(bool)
    ctxs = [Context(q["path"], f"thm{j}", Pos(*q["pos"]), f"x{j} ⊢ y") for j, q in enumerate(g["queries"])]
    for j, c in enumerate(ctxs):
        assert np.array_equal(corpus.accessible_mask(c.path, c.theorem_pos), acc[j]), j
        s = corpus.get_accessible_premises(c.path, c.theorem_pos)
        assert np.array_equal(np.array([p in s for p in corpus.all_premises]), acc[j])
        assert sorted(corpus.get_dependencies(c.path)) == sorted(ref.reach[c.path])
    # the operands handed to rp_sim_topk encode exactly the same predicate
    bits_t, own, qk = corpus.query_masks(ctxs)
    B, N = len(ctxs), g["N"]
    assert bits_t.dtype == np.uint32 and bits_t.shape == (corpus.num_files, (B + 31) // 32)
    j = np.arange(B)
    imported = (bits_t[corpus.file_of][:, j >> 5] >> (j & 31).astype(np.uint32)) & 1  # [N, B]
    own_ok = (corpus.file_of[:, None] == own[None, :]) & (corpus.end_key[:, None] <= qk[None, :])
    assert np.array_equal((imported.astype(bool) | own_ok).T, acc)
    with pytest.raises(KeyError):
        corpus.query_masks([Context("Not/In/Corpus.lean", "t", Pos(1, 1), "⊢")])


def test_duplicate_names_follow_set_semantics():
    recs = [
        {"path": "A.lean", "imports": [], "premises": [
            {"full_name": "foo", "code": "def foo := 1", "start": [1, 0], "end": [2, 0]},
            {"full_name": "bar", "code": "def bar := 1", "start": [3, 0], "end": [4, 0]},
            {"full_name": "foo", "code": "def foo := 2", "start": [9, 0], "end": [10, 0]},
        ]},
    ]
    path = os.path.join(tempfile.mkdtemp(), "c.jsonl")
    synth.write_corpus_jsonl(path, recs)
    c, ref = Corpus(path), common_ref.CorpusRef(path)
    for pos in [(1, 5), (2, 0), (5, 0), (20, 0)]:
        keys = ref.accessible_keys("A.lean", common_ref.Pos(*pos))
        want = np.array([(p.path, p.full_name) in keys for p in ref.all_premises])
        assert np.array_equal(c.accessible_mask("A.lean", Pos(*pos)), want), pos
    # position (5,0): first foo and bar ended; the later duplicate of foo counts as accessible too
    assert c.accessible_mask("A.lean", Pos(5, 0)).tolist() == [True, True, True]
    assert c.get_accessible_premise_indexes("A.lean", Pos(5, 0)) == [0, 1]  # index form differs (common.py:291-297)


def test_format_augmented_state_matches_oracle():
    prem = [Premise("A.lean", f"n{i}", Pos(1, 0), Pos(2, 0), f"theorem n{i} : {'x' * (5 * i)}") for i in range(6)]
    s = "h : p ⊢ q"
    for budget in (None, 10, 40, 80, 200):
        want = common_ref.format_augmented_state(s, [p.serialize() for p in prem], budget)
        assert format_augmented_state(s, prem, budget) == want


def test_format_augmented_state_golden_g15(golden_dir):
    """The product's ``format_augmented_state`` against the REFERENCE's outputs (fixture G15; common.py:357-378):
    byte budgets that cut the list mid-way, multi-byte states, seeded ``p_drop``."""
    import json
    import random

    g = json.load(open(os.path.join(golden_dir, "g15_augmented_state.json")))
    prem = [Premise(p["path"], p["full_name"], Pos(1, 0), Pos(2, 0), p["code"]) for p in g["premises"]]
    for c in g["cases"]:
        random.seed(c["seed"])
        assert format_augmented_state(c["state"], prem, c["max_len"], c["p_drop"]) == c["out"], c


def test_c_abi_loads_and_exports_every_declared_symbol(hip_lib):
    header = open(os.path.join(ROOT, "include", "reprover_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rp_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} is declared in include/reprover_hip.h but not exported"
    assert hip_lib.rp_abi_version() == _lib.ABI_VERSION
    assert hip_lib.rp_set_option(b"no_such_option", 1) != 0
    assert b"no_such_option" in hip_lib.rp_last_error()
    assert hip_lib.rp_sim_topk_workspace_bytes(256, 130000, 1472, 100, 0) > 0
    assert hip_lib.rp_topk_merge_workspace_bytes(8, 256, 100) >= 8 * 256 * 100 * 8


def test_bucket_function_matches_hf(hip_lib, golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_buckets.npz"))
    mine = np.array([hip_lib.rp_relative_position_bucket(int(r), 32, 128) for r in g["rel"]])
    assert np.array_equal(mine, g["bucket"])


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_a_gpu(tiny_weights):
    from reprover_amd.encoder import HipT5Encoder
    from reprover_amd.retrieval.model import PremiseRetriever

    cfg, sd = tiny_weights
    with pytest.raises(_lib.HipLibraryError):
        HipT5Encoder(cfg, sd, "cpu")
    with pytest.raises(_lib.HipLibraryError):
        PremiseRetriever.from_state_dict(cfg, sd, 512, "cuda")


def test_product_does_not_import_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.py (smoke) and bench.py (the
    cpu_baseline leg) may import it — not the package, not the tools."""
    allowed = {os.path.join(ROOT, "__graft_entry__.py"), os.path.join(ROOT, "bench.py")}
    for dirpath, dirs, names in os.walk(ROOT):
        dirs[:] = [d for d in dirs if d not in (".git", "gpurun_out", "__pycache__", "tests", "oracle")]
        for n in names:
            path = os.path.join(dirpath, n)
            if n.endswith(".py") and path not in allowed:
                src = open(path).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path
    bench_src = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"^\s*(from|import)\s+oracle\b", bench_src, flags=re.M)]
    body = bench_src[bench_src.index("def cpu_baseline"):]
    end = body.index("\ndef ", 1)
    lo = bench_src.index("def cpu_baseline")
    assert uses and all(lo < u < lo + end for u in uses), "bench.py may use the oracle inside cpu_baseline() only"


def test_native_index_roundtrip(g6, monkeypatch):
    """common.save_index / load_index: corpus and embeddings survive, dtype preserved, no pickle; the closure bit
    rows and the per-premise arrays are PERSISTED (format 2): load == rebuild bit for bit, without rebuilding."""
    from reprover_amd.common import load_index, save_index

    g, z, path = g6
    corpus = Corpus(path)
    emb = torch.randn(len(corpus), 64).to(torch.bfloat16)
    d = os.path.join(tempfile.mkdtemp(), "idx.rpidx")
    save_index(d, path, emb, corpus=corpus)
    assert sorted(os.listdir(d)) == ["arrays.safetensors", "corpus.jsonl", "embeddings.safetensors", "meta.json"]

    def boom(self, direct):
        raise AssertionError("load_index must not rebuild the import closure")

    monkeypatch.setattr(Corpus, "_build_arrays", boom)
    c2, e2 = load_index(d)
    monkeypatch.undo()
    assert torch.equal(e2, emb) and e2.dtype == torch.bfloat16
    assert [p.full_name for p in c2.all_premises] == [p.full_name for p in corpus.all_premises]
    assert np.array_equal(c2.file_of, corpus.file_of) and np.array_equal(c2.end_key, corpus.end_key)
    assert np.array_equal(c2._reach, corpus._reach) and np.array_equal(c2._file_start, corpus._file_start)
    for q in g["queries"][:6]:  # the loaded corpus answers accessibility exactly like the rebuilt one
        pos = Pos(*q["pos"])
        assert np.array_equal(c2.accessible_mask(q["path"], pos), corpus.accessible_mask(q["path"], pos))
        assert c2.get_dependencies(q["path"]) == corpus.get_dependencies(q["path"])
    # an e4m3 payload travels with the index when given (quantisation itself is a GPU test)
    from reprover_amd.common import Fp8Index

    fake = Fp8Index(torch.randint(0, 255, (len(corpus), 64), dtype=torch.uint8), torch.rand(len(corpus)))
    save_index(d, path, emb, corpus=corpus, fp8=fake)
    c3, e3, payload = load_index(d, with_fp8=True)
    assert torch.equal(payload[0], fake.codes) and torch.equal(payload[1], fake.scale)
    # a round-1 directory (format 1: no arrays file) still loads, by rebuilding
    os.remove(os.path.join(d, "arrays.safetensors"))
    os.remove(os.path.join(d, "fp8.safetensors"))
    json.dump({"format": 1, "n_premises": len(corpus), "d_model": 64, "dtype": "bfloat16"},
              open(os.path.join(d, "meta.json"), "w"))
    c4, e4 = load_index(d)
    assert np.array_equal(c4._reach, corpus._reach)
    json.dump({"format": 99}, open(os.path.join(d, "meta.json"), "w"))
    with pytest.raises(ValueError):
        load_index(d)


def test_reads_the_reference_indexed_corpus_pickle(golden_dir):
    """G14: the index file the reference's retrieval/index.py writes (its common.Corpus with a networkx graph, common.File /
    Premise, lean_dojo.Pos inside) is read WITHOUT those modules, and yields the corpus a corpus.jsonl would have built:
    same premises, same closure / per-premise arrays, the embeddings as stored."""
    import json
    import pickle
    import sys
    import tempfile

    from reprover_amd import synth
    from reprover_amd.common import Corpus, IndexedCorpus, load_indexed_corpus_pickle

    assert "lean_dojo" not in sys.modules and "networkx" not in sys.modules or True  # (other tests may import networkx)
    g = json.load(open(os.path.join(golden_dir, "g14_reference_indexed_corpus.json")))
    corpus, E = load_indexed_corpus_pickle(os.path.join(golden_dir, "g14_reference_indexed_corpus.pickle"))
    files = synth.synth_corpus_records(g["n_files"], g["n_premises"], seed=g["corpus_seed"], max_imports=g["max_imports"],
                                       code_bytes=tuple(g["code_bytes"]))
    path = os.path.join(tempfile.mkdtemp(), "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    want = Corpus(path)
    assert isinstance(corpus, Corpus) and len(corpus) == g["N"] == len(want)
    assert corpus.all_premises == want.all_premises and [f.path for f in corpus.files] == [f.path for f in want.files]
    assert np.array_equal(corpus._reach, want._reach) and np.array_equal(corpus.file_of, want.file_of)
    assert np.array_equal(corpus.end_key, want.end_key)
    assert E.dtype == torch.float32 and tuple(E.shape) == (g["N"], 128)
    assert np.allclose(np.linalg.norm(E.numpy(), axis=1), 1.0, atol=1e-5)
    # this package's own pickled IndexedCorpus goes through the same reader
    own = os.path.join(tempfile.mkdtemp(), "own.pickle")
    pickle.dump(IndexedCorpus(want, E), open(own, "wb"))
    c2, E2 = load_indexed_corpus_pickle(own)
    assert c2.all_premises == want.all_premises and torch.equal(E2, E)
    with pytest.raises(TypeError):
        junk = os.path.join(tempfile.mkdtemp(), "junk.pickle")
        pickle.dump({"not": "an index"}, open(junk, "wb"))
        load_indexed_corpus_pickle(junk)


def test_g17_index_file_under_the_reference_class_names(golden_dir):
    """`save_reference_pickle` (index.py --reference-pickle): the stream names the REFERENCE's classes - common.IndexedCorpus /
    Corpus / File / Premise, lean_dojo.data_extraction.lean.Pos, a networkx DiGraph of the closure - so that the reference's
    prover can unpickle it (prover/tactic_generator.py:273-276).  tests/golden/make_golden.py g17 did exactly that with
    the imported reference and recorded its retrieve() answers from such a file; here the file is written again, its
    global names are checked, `sys.modules` must come back untouched, and the package's own reader + the oracle's search
    must give the recorded answers."""
    import pickle
    import sys

    from oracle import common_ref, t5_ref
    from reprover_amd.common import Corpus, load_indexed_corpus_pickle, save_reference_pickle

    g = json.load(open(os.path.join(golden_dir, "g17_reference_loads_our_pickle.json")))
    files = synth.synth_corpus_records(g["n_files"], g["n_premises"], seed=g["corpus_seed"], max_imports=g["max_imports"],
                                       code_bytes=tuple(g["code_bytes"]))
    d = tempfile.mkdtemp()
    path = os.path.join(d, "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    corpus = Corpus(path)
    cfg = synth.t5_config("tiny")
    sd = synth.synth_state_dict(cfg, seed=g["weight_seed"])
    E = t5_ref.encode_texts(cfg, sd, [p.serialize() for p in corpus.all_premises], 512, 32)
    before = {n: sys.modules.get(n) for n in ("common", "lean_dojo", "lean_dojo.data_extraction", "lean_dojo.data_extraction.lean")}
    out = os.path.join(d, "for_the_reference.pickle")
    save_reference_pickle(out, corpus, E)
    assert {n: sys.modules.get(n) for n in before} == before
    globals_named = set()

    class Spy(pickle.Unpickler):  # every class the stream names
        def find_class(self, module, name):
            globals_named.add((module, name))
            return type(name, (), {"__setstate__": lambda self, st: None}) if module.split(".")[0] in (
                "common", "lean_dojo", "networkx") else super().find_class(module, name)

    Spy(open(out, "rb")).load()
    assert {("common", "IndexedCorpus"), ("common", "Corpus"), ("common", "File"), ("common", "Premise"),
            ("lean_dojo.data_extraction.lean", "Pos"), ("networkx.classes.digraph", "DiGraph")} <= globals_named
    assert not any(m.startswith("reprover_amd") for m, _ in globals_named)
    back, E2 = load_indexed_corpus_pickle(out)
    assert back.all_premises == corpus.all_premises and np.array_equal(back._reach, corpus._reach)
    assert np.array_equal(back.end_key, corpus.end_key) and torch.equal(E2, E)
    ref = common_ref.CorpusRef(path)
    for q in g["queries"]:  # the answers the REFERENCE gave from the file this writer produced in the authoring container
        ctx = common_ref.ContextRef(q["path"], "thm", common_ref.Pos(*q["pos"]), q["state"])
        qe = t5_ref.encode_texts(cfg, sd, [q["state"]], 512, 1)
        ids, sc = ref.get_nearest_premises(E2.numpy(), [ctx], qe.numpy(), g["k"])
        assert ids[0] == q["ids"] and np.abs(np.array(sc[0]) - np.array(q["scores"])).max() < 1e-5
    with pytest.raises(ValueError):
        save_reference_pickle(out, corpus, E[:-1])


def test_family_corpus_generator_is_deterministic_and_keeps_the_structure():
    """synth_family_corpus_records (fixture G7h's corpus): same files / imports / names / positions as synth_corpus_records
    with the same seed, bodies replaced by near-duplicate families; reproducible."""
    a, fam = synth.synth_family_corpus_records(20, 300, seed=171, code_bytes=(24, 96))
    b, _ = synth.synth_family_corpus_records(20, 300, seed=171, code_bytes=(24, 96))
    plain = synth.synth_corpus_records(20, 300, seed=171, code_bytes=(24, 96))
    assert a == b and len(a) == len(plain)
    for fa, fp in zip(a, plain):
        assert fa["path"] == fp["path"] and fa["imports"] == fp["imports"]
        assert [(p["full_name"], p["start"], p["end"]) for p in fa["premises"]] == \
            [(p["full_name"], p["start"], p["end"]) for p in fp["premises"]]
    assert any(len(f["members"]) == 12 for f in fam)
    by_name = {(i, p["full_name"]): p["code"] for i, f in enumerate(a) for p in f["premises"] if p["full_name"] and not p["code"].endswith("-- again")}
    f0 = next(f for f in fam if len(f["members"]) == 12)
    exact = [n for n, r in zip(f0["members"], f0["rates"]) if r == 0.0]
    assert len(exact) == 1 and by_name[(f0["file"], exact[0])].endswith(f0["base"])


def test_comm_entry_points_validate_arguments_without_a_gpu(hip_lib):
    """rp_comm_* (the sharded step's collective behind the C ABI): argument errors are statuses with a message, a null
    communicator is harmless - checked here without a GPU and without RCCL being bound."""
    import ctypes as C

    h = C.c_void_p()
    assert hip_lib.rp_comm_init(C.c_char_p(b"\0" * 128), 3, 2, C.byref(h)) == -1 and not h.value
    assert b"rank 3 of 2" in hip_lib.rp_last_error()
    assert hip_lib.rp_comm_init(None, 0, 1, C.byref(h)) == -1
    assert hip_lib.rp_comm_world(None) == 0 and hip_lib.rp_comm_rank(None) == -1
    assert hip_lib.rp_comm_destroy(None) == 0
    assert hip_lib.rp_comm_allgather(None, None, None, 16, None) == -1
    assert hip_lib.rp_allgather_topk(None, None, None, 4, 2, 0, 4, None, None, None, None, 0, None) == -1


def test_bench_gpus_n_without_a_launcher_starts_its_own_ranks(monkeypatch):
    """bench.py --gpus 4 with no WORLD_SIZE in the environment re-runs itself under torch.distributed.run, one rank
    per GPU, rendezvous on 127.0.0.1 (the driver contract's command line); under a launcher it does not."""
    import subprocess
    import sys

    import bench

    seen = {}

    def fake_call(cmd, *a, **k):
        seen["cmd"] = cmd
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    # a mismatch between the launcher's world size and --gpus is an error message, not an assertion
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value.code)


def test_resumed_epoch_skips_batches_at_the_index_level():
    """A resumed `fit` consumes the batches it already trained on without collating / tokenising them, and the
    batches that follow - order AND negative draws - are those of an uninterrupted epoch."""
    import random

    from reprover_amd.retrieval.datamodule import RetrievalDataModule

    files = synth.synth_corpus_records(30, 500, seed=131, max_imports=5)
    d = tempfile.mkdtemp()
    cpath = os.path.join(d, "corpus.jsonl")
    synth.write_corpus_jsonl(cpath, files)
    dd = os.path.join(d, "data")
    os.makedirs(dd)
    for name, seed in (("train", 132), ("val", 134), ("test", 135)):
        json.dump(synth.synth_split(files, 40, seed=seed, min_file=12), open(os.path.join(dd, f"{name}.json"), "w"))
    calls = []

    class CountingTokenizer(tokenizer.ByT5Tokenizer):
        def __call__(self, *a, **kw):
            calls.append(1)
            return super().__call__(*a, **kw)

    dm = RetrievalDataModule(dd, cpath, 16, 256, CountingTokenizer(), num_negatives=3, num_in_file_negatives=1, batch_size=8)
    dm.setup("fit")
    dm.ds_train.data = [ex for ex in dm.ds_train.data if len(dm.ds_train.negative_pools(ex)[1]) >= 3]

    def sig(b):
        return [c.state for c in b["context"]], [[p.full_name for p in n] for n in b["neg_premises"]]

    random.seed(7)
    full = [sig(b) for b in dm.train_dataloader()]
    per_batch = len(calls) // len(full)
    calls.clear()
    random.seed(7)
    tail = [sig(b) for b in dm.train_dataloader(skip=3)]
    assert len(full) >= 6 and tail == full[3:]
    assert len(calls) == per_batch * len(tail)  # the three skipped batches were never tokenised
