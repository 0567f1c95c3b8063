"""Evaluation-side data loading of the retriever — the inference half of the reference's
``retrieval/datamodule.py`` (lean-dojo/ReProver): one example per traced tactic
(``RetrievalDataset._load_data`` with ``is_train=False``, datamodule.py:44-90), context tokenisation
in ``collate`` (:130-144), and the val / predict splits (:234-267).  Of the training half only the
deterministic part is here: ``collate_train`` (the positive / negative tokenisation and the label matrix of
datamodule.py:146-191), which feeds ``PremiseRetriever.forward``; the random negative sampling of
``__getitem__`` (:95-128) is not.
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, Iterator, List, Optional

from ..common import Context, Corpus, Pos, get_all_pos_premises

Example = Dict[str, Any]
Batch = Dict[str, Any]


class RetrievalDataset:
    def __init__(self, data_paths: List[str], corpus: Corpus, max_seq_len: int, tokenizer, is_train: bool = False):
        assert not is_train, "the training branch (negatives, labels) is outside the retrieval hot path"
        self.corpus = corpus
        self.max_seq_len = max_seq_len
        self.tokenizer = tokenizer
        self.is_train = False
        self.data: List[Example] = []
        for path in data_paths:
            self.data.extend(self._load_data(path))

    def _load_data(self, data_path: str) -> List[Example]:
        data = []
        with open(data_path) as fh:
            theorems = json.load(fh)
        for thm in theorems:
            file_path = thm["file_path"]
            for i, tac in enumerate(thm["traced_tactics"]):
                context = Context(file_path, thm["full_name"], Pos(*thm["start"]), tac["state_before"])
                data.append(
                    {
                        "url": thm["url"],
                        "commit": thm["commit"],
                        "file_path": thm["file_path"],
                        "full_name": thm["full_name"],
                        "start": thm["start"],
                        "tactic_idx": i,
                        "context": context,
                        "all_pos_premises": get_all_pos_premises(tac["annotated_tactic"], self.corpus),
                    }
                )
        return data

    def __len__(self) -> int:
        return len(self.data)

    def __getitem__(self, idx: int) -> Example:
        return self.data[idx]

    def collate(self, examples: List[Example]) -> Batch:
        context = [ex["context"] for ex in examples]
        tok = self.tokenizer(
            [c.serialize() for c in context], padding="longest", max_length=self.max_seq_len, truncation=True,
            return_tensors="pt",
        )
        batch: Batch = {"context": context, "context_ids": tok.input_ids, "context_mask": tok.attention_mask}
        for k in examples[0].keys():
            if k not in batch:
                batch[k] = [ex[k] for ex in examples]
        return batch

    def batches(self, batch_size: int) -> Iterator[Batch]:
        """In-order, drop_last=False — what the reference's eval DataLoaders yield."""
        for i in range(0, len(self.data), batch_size):
            yield self.collate(self.data[i : i + batch_size])


def label_matrix(examples: List[Example], num_negatives: int):
    """``label[j, k] = 1`` iff column k's premise is one of example j's positive premises, columns = the batch's
    positives followed by negative list 0, 1, ... (datamodule.py:160-175)."""
    import torch

    n = len(examples)
    label = torch.zeros(n, n * (1 + num_negatives))
    for j in range(n):
        all_pos = examples[j]["all_pos_premises"]
        for k in range(n * (1 + num_negatives)):
            prem = examples[k]["pos_premise"] if k < n else examples[k % n]["neg_premises"][k // n - 1]
            label[j, k] = float(prem in all_pos)
    return label


def collate_train(examples: List[Example], tokenizer, max_seq_len: int, num_negatives: int) -> Batch:
    """The reference's ``collate`` with ``is_train=True`` (datamodule.py:130-198) for examples that already carry
    ``pos_premise`` and ``neg_premises``: tokenised contexts / positives / negative lists + the label matrix."""
    def tok(texts):
        return tokenizer(texts, padding="longest", max_length=max_seq_len, truncation=True, return_tensors="pt")

    context = [ex["context"] for ex in examples]
    t = tok([c.serialize() for c in context])
    batch: Batch = {"context": context, "context_ids": t.input_ids, "context_mask": t.attention_mask}
    pos = [ex["pos_premise"] for ex in examples]
    t = tok([p.serialize() for p in pos])
    batch.update(pos_premise=pos, pos_premise_ids=t.input_ids, pos_premise_mask=t.attention_mask,
                 label=label_matrix(examples, num_negatives), neg_premises=[], neg_premises_ids=[], neg_premises_mask=[])
    for i in range(num_negatives):
        neg = [ex["neg_premises"][i] for ex in examples]
        t = tok([p.serialize() for p in neg])
        batch["neg_premises"].append(neg)
        batch["neg_premises_ids"].append(t.input_ids)
        batch["neg_premises_mask"].append(t.attention_mask)
    for k in examples[0].keys():
        if k not in batch:
            batch[k] = [ex[k] for ex in examples]
    return batch


class RetrievalDataModule:
    """``data_path`` holds ``{train,val,test}.json``; ``corpus_path`` is ``corpus.jsonl``
    (datamodule.py:201-228).  Only the val and predict splits are built."""

    def __init__(self, data_path: str, corpus_path: str, eval_batch_size: int, max_seq_len: int, tokenizer,
                 corpus: Optional[Corpus] = None, **_ignored_training_args) -> None:
        self.data_path = data_path
        self.eval_batch_size = eval_batch_size
        self.max_seq_len = max_seq_len
        self.tokenizer = tokenizer
        self.corpus = corpus if corpus is not None else Corpus(corpus_path)
        self.ds_val: Optional[RetrievalDataset] = None
        self.ds_pred: Optional[RetrievalDataset] = None

    def setup(self, stage: Optional[str] = None) -> None:
        def split(name):
            return os.path.join(self.data_path, f"{name}.json")

        if stage in (None, "fit", "validate"):
            self.ds_val = RetrievalDataset([split("val")], self.corpus, self.max_seq_len, self.tokenizer)
        if stage in (None, "fit", "predict"):
            self.ds_pred = RetrievalDataset([split(s) for s in ("train", "val", "test")], self.corpus,
                                            self.max_seq_len, self.tokenizer)

    def val_dataloader(self) -> Iterator[Batch]:
        return self.ds_val.batches(self.eval_batch_size)

    def predict_dataloader(self) -> Iterator[Batch]:
        return self.ds_pred.batches(self.eval_batch_size)
