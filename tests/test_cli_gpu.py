"""The CLI-level drop-ins on the GPU with a tiny checkpoint written in HuggingFace layout:
load_hf (config.json + model.safetensors), retrieval/index.py (same flags as the reference),
predict / validate drivers, evaluate.py."""
import json
import os
import pickle
import tempfile

import numpy as np
import pytest
import torch
import yaml
from safetensors.torch import save_file

from oracle import common_ref, eval_ref, t5_ref
from reprover_amd import synth
from reprover_amd.common import IndexedCorpus
from reprover_amd.retrieval import evaluate as evaluate_cli
from reprover_amd.retrieval import index as index_cli
from reprover_amd.retrieval import main as main_cli
from reprover_amd.retrieval.model import PremiseRetriever

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workdir(tiny_weights):
    cfg, sd = tiny_weights
    d = tempfile.mkdtemp()
    ckpt = os.path.join(d, "ckpt")
    os.makedirs(ckpt)
    hf_cfg = {k: cfg[k] for k in ("vocab_size", "d_model", "d_kv", "num_heads", "d_ff", "num_layers",
                                  "relative_attention_num_buckets", "relative_attention_max_distance",
                                  "layer_norm_epsilon", "feed_forward_proj")}
    json.dump(hf_cfg, open(os.path.join(ckpt, "config.json"), "w"))
    save_file({k: v.clone().contiguous() for k, v in sd.items() if k != "encoder.embed_tokens.weight"},
              os.path.join(ckpt, "model.safetensors"))
    files = synth.synth_corpus_records(30, 500, seed=91, max_imports=6)
    cpath = os.path.join(d, "corpus.jsonl")
    synth.write_corpus_jsonl(cpath, files)
    sdir = os.path.join(d, "split")
    os.makedirs(sdir)
    from reprover_amd.common import Corpus, Pos

    corpus = Corpus(cpath)

    def enough(path, start):  # the reference raises ValueError below k accessible premises
        return corpus.accessible_mask(path, Pos(*start)).sum() >= 12

    splits = {"train": synth.synth_split(files, 12, seed=92, min_file=15, accept=enough),
              "val": synth.synth_split(files, 10, seed=93, min_file=15, accept=enough),
              "test": synth.synth_split(files, 8, seed=94, min_file=15, accept=enough)}
    for name, sp in splits.items():
        json.dump(sp, open(os.path.join(sdir, f"{name}.json"), "w"))
    conf = {"model": {"model_name": ckpt, "num_retrieved": 10},
            "data": {"data_path": sdir, "corpus_path": cpath, "eval_batch_size": 16, "max_seq_len": 256}}
    yaml.safe_dump(conf, open(os.path.join(d, "conf.yaml"), "w"))
    return d, ckpt, cpath, sdir, splits, cfg, sd


def test_index_cli_writes_reference_style_pickle(workdir):
    d, ckpt, cpath, sdir, splits, cfg, sd = workdir
    out = os.path.join(d, "indexed.pickle")
    index_cli.main(["--ckpt_path", ckpt, "--corpus-path", cpath, "--output-path", out, "--batch-size", "32"])
    ic = pickle.load(open(out, "rb"))
    assert isinstance(ic, IndexedCorpus) and ic.embeddings.dtype == torch.float32 and ic.embeddings.device.type == "cpu"
    assert ic.embeddings.shape == (len(ic.corpus), cfg["d_model"])
    texts = [p.serialize() for p in common_ref.CorpusRef(cpath).all_premises]
    ref = t5_ref.encode_texts(cfg, sd, texts[:64], 2048, 16)
    cos = torch.nn.functional.cosine_similarity(ic.embeddings[:64], ref, dim=1)
    assert cos.min().item() > 0.999
    with pytest.raises(FileExistsError):
        PremiseRetriever.load_hf(os.path.join(d, "no_such_ckpt"), 256, "cuda:0")


def test_index_cli_reference_pickle_flag(workdir):
    """`index.py --reference-pickle`: the same index under the REFERENCE's class names (common.IndexedCorpus / Corpus / File /
    Premise, lean_dojo.data_extraction.lean.Pos, a networkx closure graph), the file the reference's prover unpickles
    (prover/tactic_generator.py:273-276; verified against the imported reference by fixture G17).  Here: the CLI writes it, the
    stream names no class of this package, and it reads back to the index the plain flag-less run writes."""
    d, ckpt, cpath, sdir, splits, cfg, sd = workdir
    plain, ref = os.path.join(d, "plain.pickle"), os.path.join(d, "for_reference.pickle")
    index_cli.main(["--ckpt_path", ckpt, "--corpus-path", cpath, "--output-path", plain, "--batch-size", "32"])
    index_cli.main(["--ckpt_path", ckpt, "--corpus-path", cpath, "--output-path", ref, "--batch-size", "32", "--reference-pickle"])
    named = set()

    class Spy(pickle.Unpickler):
        def find_class(self, module, name):
            named.add((module, name))
            return type(name, (), {"__setstate__": lambda self, st: None}) if module.split(".")[0] in (
                "common", "lean_dojo", "networkx") else super().find_class(module, name)

    Spy(open(ref, "rb")).load()
    assert ("common", "IndexedCorpus") in named and ("lean_dojo.data_extraction.lean", "Pos") in named
    assert not any(m.startswith("reprover_amd") for m, _ in named)
    from reprover_amd.common import load_indexed_corpus_pickle

    c1, e1 = load_indexed_corpus_pickle(plain)
    c2, e2 = load_indexed_corpus_pickle(ref)
    assert c1.all_premises == c2.all_premises and torch.equal(e1, e2)
    m = PremiseRetriever.load_hf(ckpt, 256, "cuda:0")
    m.load_corpus(ref)  # and the retriever takes it like any indexed corpus
    assert not m.embeddings_staled and len(m.corpus) == len(c1)


def test_predict_validate_and_evaluate(workdir):
    d, ckpt, cpath, sdir, splits, cfg, sd = workdir
    log_dir = os.path.join(d, "logs")
    main_cli.main(["predict", "--config", os.path.join(d, "conf.yaml"), "--log-dir", log_dir])
    preds = pickle.load(open(os.path.join(log_dir, "predictions.pickle"), "rb"))
    n_examples = sum(len(t["traced_tactics"]) for sp in splits.values() for t in sp)
    assert len(preds) == n_examples and all(len(p["retrieved_premises"]) == 10 for p in preds)
    # every retrieved premise is accessible and scores are sorted
    oc = common_ref.CorpusRef(cpath)
    for p in preds[:20]:
        keys = oc.accessible_keys(p["context"].path, common_ref.Pos(*p["context"].theorem_pos))
        assert all((q.path, q.full_name) in keys for q in p["retrieved_premises"])
        assert all(a >= b for a, b in zip(p["scores"], p["scores"][1:]))
    # evaluate.py on the pickle == the oracle's metrics on the same retrieved lists
    pm = evaluate_cli.load_preds_map(os.path.join(log_dir, "predictions.pickle"))
    for name, sp in splits.items():
        r = evaluate_cli._eval(sp, pm)
        ex = eval_ref.load_eval_examples(os.path.join(sdir, f"{name}.json"), oc)
        where = {(q.path, q.full_name, tuple(q.start)): i for i, q in enumerate(oc.all_premises)}
        retrieved = [[where[(q.path, q.full_name, tuple(q.start))] for q in
                      pm[(e["file_path"], e["full_name"], tuple(e["start"]), e["tactic_idx"])]["retrieved_premises"]]
                     for e in ex]
        assert np.allclose(r, eval_ref.eval_predictions(ex, retrieved), atol=1e-9)
    # validate: Recall@k / MRR as validation_step logs them
    model = PremiseRetriever.load_hf(ckpt, 256, "cuda:0")
    model.num_retrieved = 10
    from reprover_amd.retrieval.datamodule import RetrievalDataModule

    dm = RetrievalDataModule(sdir, cpath, 16, 256, model.tokenizer)
    m = main_cli.run_validate(model, dm)
    ex = eval_ref.load_eval_examples(os.path.join(sdir, "val.json"), oc)
    keymap = {(e["file_path"], e["full_name"], tuple(e["start"]), e["tactic_idx"]): i for i, e in enumerate(ex)}
    retrieved = [None] * len(ex)
    for key, p in pm.items():
        if key in keymap:
            retrieved[keymap[key]] = [where[(q.path, q.full_name, tuple(q.start))] for q in p["retrieved_premises"]]
    rec, mrr = eval_ref.validation_metrics([e["all_pos_premises"] for e in ex], retrieved, 10)
    assert abs(m["MRR"] - mrr) < 1e-9 and np.allclose([m[f"Recall@{j + 1}_val"] for j in range(10)], rec, atol=1e-9)


def test_on_predict_start_reads_an_attached_trainer(workdir):
    """The reference's ``on_predict_start(self)`` takes no arguments and reads ``self.trainer.datamodule`` (model.py:274-279):
    a caller that attaches a Lightning-style trainer object gets the same behaviour here."""
    import types

    from reprover_amd.retrieval.datamodule import RetrievalDataModule

    d, ckpt, cpath, sdir, splits, cfg, sd = workdir
    model = PremiseRetriever.load_hf(ckpt, 256, "cuda:0")
    dm = RetrievalDataModule(sdir, cpath, 16, 256, model.tokenizer)
    with pytest.raises(TypeError):
        model.on_predict_start()
    model.trainer = types.SimpleNamespace(datamodule=dm)
    model.on_predict_start()
    assert model.corpus is dm.corpus and not model.embeddings_staled
    assert model.corpus_embeddings.shape == (len(dm.corpus), cfg["d_model"]) and model.predict_step_outputs == []


def test_native_index_directory_via_cli(workdir):
    d, ckpt, cpath, sdir, splits, cfg, sd = workdir
    out = os.path.join(d, "native.rpidx")
    index_cli.main(["--ckpt_path", ckpt, "--corpus-path", cpath, "--output-path", out])
    assert sorted(os.listdir(out)) == ["arrays.safetensors", "corpus.jsonl", "embeddings.safetensors", "meta.json"]
    m = PremiseRetriever.load_hf(ckpt, 256, "cuda:0")
    m.load_corpus(out)
    assert not m.embeddings_staled and m.corpus_embeddings.dtype == torch.bfloat16
    # with RP_INDEX_FP8=1 the e4m3 form is persisted too and a retriever asking for it loads it as is
    out8 = os.path.join(d, "native8.rpidx")
    os.environ["RP_INDEX_FP8"] = "1"
    try:
        index_cli.main(["--ckpt_path", ckpt, "--corpus-path", cpath, "--output-path", out8])
    finally:
        del os.environ["RP_INDEX_FP8"]
    assert "fp8.safetensors" in os.listdir(out8)
    m8 = PremiseRetriever(ckpt, max_seq_len=256, device="cuda:0", index_dtype="fp8")
    m8.load_corpus(out8)
    from reprover_amd.common import Fp8Index

    want = Fp8Index.quantize(m8.corpus_embeddings, m8.device)
    assert m8._fp8_index is not None and torch.equal(m8._fp8_index.codes, want.codes) and torch.equal(
        m8._fp8_index.scale, want.scale)
    ref = PremiseRetriever.load_hf(ckpt, 256, "cuda:0")
    ref.load_corpus(cpath)
    ex = splits["val"][0]
    state = ex["traced_tactics"][0]["state_before"]
    from reprover_amd.common import Pos

    a = m.retrieve(state, ex["file_path"], ex["full_name"], Pos(*ex["start"]), 10)
    b = ref.retrieve(state, ex["file_path"], ex["full_name"], Pos(*ex["start"]), 10)
    assert [p.full_name for p in a[0]] == [p.full_name for p in b[0]] and a[1] == b[1]


def test_index_cli_two_ranks_equals_one(workdir):
    """BASELINE configs[3] (re-index at 1 vs N GPUs), functionally: two ranks under torch.distributed.run
    (both on this box's one GPU, gloo for the gather — RCCL refuses two ranks per device) must write the
    same index a single process writes, bit for bit."""
    import subprocess
    import sys

    d, ckpt, cpath, sdir, splits, cfg, sd = workdir
    one = os.path.join(d, "one.pickle")
    two = os.path.join(d, "two.pickle")
    index_cli.main(["--ckpt_path", ckpt, "--corpus-path", cpath, "--output-path", one])
    env = dict(os.environ, RP_DIST_SHARE_GPU="1", RP_DIST_BACKEND="gloo",
               PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    port = 29700 + os.getpid() % 1000
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), "-m",
                          "reprover_amd.retrieval.index", "--ckpt_path", ckpt, "--corpus-path", cpath,
                          "--output-path", two], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    a, b = pickle.load(open(one, "rb")), pickle.load(open(two, "rb"))
    assert len(a.corpus) == len(b.corpus) and torch.equal(a.embeddings, b.embeddings)


def test_predict_two_ranks_sharded_index_equals_one(workdir):
    """BASELINE configs[2] through the product driver: two ranks, row-sharded index, per-rank top-k
    merged through one all-gather — the predictions must be those of the single-process run (same
    premises, same scores bit for bit: masked top-k is a decomposable reduction)."""
    import subprocess
    import sys

    d, ckpt, cpath, sdir, splits, cfg, sd = workdir
    log1, log2 = os.path.join(d, "logs_one"), os.path.join(d, "logs_two")
    main_cli.main(["predict", "--config", os.path.join(d, "conf.yaml"), "--log-dir", log1])
    env = dict(os.environ, RP_DIST_SHARE_GPU="1", RP_DIST_BACKEND="gloo",
               PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    port = 29800 + os.getpid() % 1000
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), "-m",
                          "reprover_amd.retrieval.main", "predict", "--config", os.path.join(d, "conf.yaml"),
                          "--log-dir", log2], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    a = pickle.load(open(os.path.join(log1, "predictions.pickle"), "rb"))
    b = pickle.load(open(os.path.join(log2, "predictions.pickle"), "rb"))
    assert len(a) == len(b) > 0
    for x, y in zip(a, b):
        assert [(p.path, p.full_name, tuple(p.start)) for p in x["retrieved_premises"]] == \
               [(p.path, p.full_name, tuple(p.start)) for p in y["retrieved_premises"]]
        assert x["scores"] == y["scores"]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's scaling command when it does not
    wrap the script in torch.distributed.run) starts its two ranks itself.  Here both ranks share this box's one GPU and
    gather over gloo (a functional run of the N > 1 step, not a measurement): ONE JSON line, two collectives per step,
    and the sharded search + merge equals the single-GPU answer."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RP_BENCH_SHARE_GPU="1", RP_BENCH_BACKEND="gloo")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--headline-only"], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["sharded_merge_equals_single_gpu"] is True
    assert d["config"]["collectives_per_step"] == 2
    assert d["config"]["all_counts_eq_k"] is True
    # the top-k stage timed WITH its collective, under both exchange forms (packed all-gather = the default and
    # north_star's form; all-to-all of per-destination slices): both must reproduce the single-GPU lists
    assert d["config"]["exchange"].startswith("allgather")  # auto: by bytes, the all-gather below 4 ranks
    t = d["topk_only"]
    for how in ("allgather", "alltoall"):
        assert t[how]["sharded_merge_equals_single_gpu"] is True, how
        assert t[how]["qps"] > 0 and t[how]["exchange_us"] > 0
    blk = 2 * 256 * (2 * 100 + 1) * 4
    assert t["allgather"]["collective_bytes_received_per_rank"] == blk
    assert t["alltoall"]["collective_bytes_received_per_rank"] == blk // 2
    # ... and the step itself on the sliced exchange
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--headline-only", "--exchange", "alltoall"], env=env, capture_output=True, text=True, timeout=900,
                         cwd=root)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert d["config"]["exchange"] == "alltoall" and d["config"]["collectives_per_step"] == 2
    assert d["config"]["sharded_merge_equals_single_gpu"] is True and d["config"]["all_counts_eq_k"] is True


def test_load_lightning_checkpoint_file(workdir, tmp_path):
    """PremiseRetriever.load(ckpt_path, device, freeze) (reference retrieval/model.py:48-50 -> common.py:414-425): a
    PyTorch-Lightning checkpoint FILE - ``state_dict`` under the ``encoder.`` attribute prefix + ``hyper_parameters`` as
    ``save_hyperparameters()`` records them - gives the retriever the HF directory of the same weights gives, bit for bit;
    the geometry comes from the tensor shapes when ``model_name`` is not a local directory.  (Written here in Lightning's
    layout: pytorch_lightning itself is not in this image, so the file cannot come from the reference's own trainer.)"""
    d, ckpt, cpath, sdir, splits, cfg, sd = workdir
    state = {"encoder." + k: v.clone() for k, v in sd.items()}
    state["some_metric.total"] = torch.zeros(())  # non-encoder entries are ignored (strict=False upstream)
    path = str(tmp_path / "last.ckpt")
    torch.save({"epoch": 3, "global_step": 1000, "pytorch-lightning_version": "2.0.0", "state_dict": state,
                "hyper_parameters": {"model_name": "google/byt5-small", "lr": 1e-4, "warmup_steps": 2000,
                                     "max_seq_len": 256, "num_retrieved": 10}}, path)
    a = PremiseRetriever.load(path, "cuda:0", freeze=True)
    b = PremiseRetriever.load_hf(ckpt, 256, "cuda:0")
    assert (a.max_seq_len, a.num_retrieved, a.lr, a.warmup_steps) == (256, 10, 1e-4, 2000)
    for key in ("d_model", "num_heads", "num_layers", "d_ff", "vocab_size"):
        assert a.encoder.cfg[key] == cfg[key], key
    texts = ["theorem foo (a b : Nat) : a + b = b + a", "x"]
    assert torch.equal(a.encode_texts(texts), b.encode_texts(texts))
    with pytest.raises(RuntimeError, match="freeze"):
        a.training_step({})
    with pytest.raises(FileExistsError):
        PremiseRetriever.load(str(tmp_path / "missing.ckpt"), "cuda:0", freeze=False)
    ds = tmp_path / "zero_ckpt"
    ds.mkdir()
    (ds / "zero_to_fp32.py").write_text("# DeepSpeed's conversion script lives here")
    with pytest.raises(NotImplementedError, match="DeepSpeed"):
        PremiseRetriever.load(str(ds), "cuda:0", freeze=False)
    # a HuggingFace directory goes the load_hf way
    c = PremiseRetriever.load(ckpt, "cuda:0", freeze=False)
    assert torch.equal(c.encode_texts(texts), b.encode_texts(texts))


def test_fit_cli_keeps_a_checkpoint_without_a_log_dir(workdir, tmp_path, monkeypatch):
    """`fit` from a reference-style YAML that names no root directory (Lightning's Trainer then defaults
    default_root_dir to the working directory and always keeps a checkpoint): the weights must not be discarded
    (ADVICE r05).  And a resume whose directory is missing falls back to the `.old` copy a crash between the two
    renames of the checkpoint swap leaves behind."""
    import yaml

    d, ckpt, cpath, sdir, splits, cfg, sd = workdir
    conf = {"seed_everything": 3407, "trainer": {"max_steps": 2, "gradient_clip_val": 1.0},
            "model": {"model_name": ckpt, "num_retrieved": 10, "lr": 1e-4, "warmup_steps": 1},
            "data": {"data_path": sdir, "corpus_path": cpath, "eval_batch_size": 16, "max_seq_len": 256,
                     "batch_size": 4, "num_negatives": 1, "num_in_file_negatives": 0}}
    cpath_yaml = str(tmp_path / "fit.yaml")
    yaml.safe_dump(conf, open(cpath_yaml, "w"))
    monkeypatch.chdir(tmp_path)
    main_cli.main(["fit", "--config", cpath_yaml])
    ck = tmp_path / "lightning_logs" / "checkpoint"
    assert (ck / "training_state.safetensors").exists() and (ck / "loop_state.json").exists()
    reloaded = PremiseRetriever.load_hf(str(ck), 256, "cuda:0")  # an HF directory of the trained weights
    base = PremiseRetriever.load_hf(ckpt, 256, "cuda:0")
    texts = ["theorem foo : a = b"]
    assert not torch.equal(reloaded.encode_texts(texts), base.encode_texts(texts))
    os.replace(ck, str(ck) + ".old")  # what a crash inside save_fit_checkpoint's swap leaves
    main_cli.main(["fit", "--config", cpath_yaml, "--max-steps", "3", "--resume-from", str(ck), "--log-dir", str(tmp_path / "second")])
    import json as _json
    assert _json.load(open(tmp_path / "second" / "checkpoint" / "loop_state.json"))["step"] == 3


def test_validation_hooks_on_the_class(workdir):
    """on_validation_start / validation_step as methods of PremiseRetriever (reference retrieval/model.py:212-268), called
    the way Lightning calls them, give the epoch metrics run_validate reports (the oracle's Recall@k / MRR)."""
    from reprover_amd.retrieval.datamodule import RetrievalDataModule

    d, ckpt, cpath, sdir, splits, cfg, sd = workdir
    model = PremiseRetriever.load_hf(ckpt, 256, "cuda:0")
    model.num_retrieved = 10
    dm = RetrievalDataModule(sdir, cpath, 16, 256, model.tokenizer)
    dm.setup("validate")
    model.load_corpus(dm.corpus)

    class _Trainer:  # the attribute Lightning sets; only datamodule.eval_batch_size is read
        datamodule = dm

    model.trainer = _Trainer()
    model.on_validation_start()
    assert not model.embeddings_staled
    for i, batch in enumerate(dm.val_dataloader()):
        assert model.validation_step(batch, i) is None
    got = model.epoch_metrics()
    want = main_cli.run_validate(model, dm)
    assert set(f"Recall@{j + 1}_val" for j in range(10)) <= set(got) and "MRR" in got
    for k_, v_ in got.items():
        assert abs(v_ - want[k_]) < 1e-12, k_
