"""Evaluation-side host logic (datamodule eval branch, evaluate.py metrics, validation metrics)
against the golden produced by the reference's own RetrievalDataset / _eval (tests/golden/g8)."""
import json
import os
import tempfile

import numpy as np
import pytest

from oracle import common_ref, eval_ref
from reprover_amd import synth, tokenizer
from reprover_amd.common import Corpus
from reprover_amd.retrieval.datamodule import RetrievalDataModule, RetrievalDataset
from reprover_amd.retrieval.evaluate import _eval, recall_and_mrr


@pytest.fixture(scope="module")
def g8(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "g8_eval.json")))
    files = synth.synth_corpus_records(g["n_files"], g["n_premises"], seed=g["corpus_seed"], max_imports=g["max_imports"])
    d = tempfile.mkdtemp()
    cpath = os.path.join(d, "corpus.jsonl")
    synth.write_corpus_jsonl(cpath, files)
    os.makedirs(os.path.join(d, "split"))
    split = synth.synth_split(files, g["n_theorems"], seed=g["split_seed"], min_file=g["min_file"])
    for name in ("train", "val", "test"):
        json.dump(split if name == "val" else [], open(os.path.join(d, "split", f"{name}.json"), "w"))
    return g, cpath, os.path.join(d, "split"), split


def test_oracle_eval_examples_and_metrics(g8):
    g, cpath, sdir, split = g8
    oc = common_ref.CorpusRef(cpath)
    assert eval_ref.load_eval_examples(os.path.join(sdir, "val.json"), oc) == g["examples"]
    r = eval_ref.eval_predictions(g["examples"], g["retrieved"])
    assert np.allclose(r, (g["R1"], g["R10"], g["MRR"]), atol=1e-9)
    rec, mrr = eval_ref.validation_metrics([e["all_pos_premises"] for e in g["examples"]], g["retrieved"], 20)
    assert np.allclose(rec, g["recall_at_k"], atol=1e-9) and abs(mrr - g["MRR"]) < 1e-12


def test_product_dataset_collate_and_metrics(g8):
    g, cpath, sdir, split = g8
    corpus = Corpus(cpath)
    tok = tokenizer.ByT5Tokenizer()
    ds = RetrievalDataset([os.path.join(sdir, "val.json")], corpus, 256, tok)
    where = {id(p): i for i, p in enumerate(corpus.all_premises)}
    got = [{"file_path": e["file_path"], "full_name": e["full_name"], "start": e["start"], "tactic_idx": e["tactic_idx"],
            "state": e["context"].state, "all_pos_premises": sorted(where[id(p)] for p in e["all_pos_premises"])}
           for e in ds.data]
    assert got == g["examples"]
    batch = ds.collate(ds.data[:7])
    assert sorted(batch) == g["collate_keys"]
    assert batch["context_ids"].tolist() == g["collate_ids_first7"]
    assert batch["context_mask"].sum().item() == sum(len(e["context"].state.encode()) + 1 for e in ds.data[:7])
    # datamodule: predict split = train + val + test, in order, drop_last False
    dm = RetrievalDataModule(sdir, cpath, eval_batch_size=16, max_seq_len=256, tokenizer=tok, corpus=corpus)
    dm.setup("predict")
    sizes = [len(b["context"]) for b in dm.predict_dataloader()]
    assert sum(sizes) == len(g["examples"]) and all(s == 16 for s in sizes[:-1])
    # metrics on the golden's synthetic predictions
    preds = []
    for e, ret in zip(ds.data, g["retrieved"]):
        preds.append({**{k: e[k] for k in ("file_path", "full_name", "start", "tactic_idx", "all_pos_premises")},
                      "retrieved_premises": [corpus.all_premises[i] for i in ret]})
    pm = {(p["file_path"], p["full_name"], tuple(p["start"]), p["tactic_idx"]): p for p in preds}
    assert np.allclose(_eval(split, pm), (g["R1"], g["R10"], g["MRR"]), atol=1e-9)
    rec, mrr, n = recall_and_mrr([p["all_pos_premises"] for p in preds], [p["retrieved_premises"] for p in preds], 20)
    assert np.allclose(rec, g["recall_at_k"], atol=1e-9) and abs(mrr - g["MRR"]) < 1e-12
    assert n == sum(1 for e in g["examples"] if e["all_pos_premises"])
