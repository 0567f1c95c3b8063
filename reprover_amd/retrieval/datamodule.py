"""Data loading of the retriever - the host side of the reference's ``retrieval/datamodule.py`` (lean-dojo/ReProver):
one example per traced tactic for evaluation / prediction (``_load_data`` with ``is_train=False``, datamodule.py:44-90),
one per (tactic, positive premise) for training (:60-75); negative sampling for training examples (``__getitem__``,
:95-128: in-file negatives first, the rest from the other accessible files); context tokenisation (``collate``,
:130-144) and the training collate (:146-191: positives, negative lists, label matrix), which feeds
``PremiseRetriever.forward`` / ``training_step``; the train / val / predict splits (:234-267).
"""
from __future__ import annotations

import json
import os
import random
from typing import Any, Dict, Iterator, List, Optional, Tuple

from ..common import Context, Corpus, Pos, get_all_pos_premises

Example = Dict[str, Any]
Batch = Dict[str, Any]


class RetrievalDataset:
    def __init__(self, data_paths: List[str], corpus: Corpus, max_seq_len: int, tokenizer, is_train: bool = False,
                 num_negatives: int = 0, num_in_file_negatives: int = 0):
        assert 0 <= num_in_file_negatives <= num_negatives or not is_train
        self.corpus = corpus
        self.max_seq_len = max_seq_len
        self.tokenizer = tokenizer
        self.is_train = is_train
        self.num_negatives = num_negatives
        self.num_in_file_negatives = num_in_file_negatives
        self.data: List[Example] = []
        for path in data_paths:
            self.data.extend(self._load_data(path))

    def _load_data(self, data_path: str) -> List[Example]:
        with open(data_path) as fh:
            theorems = json.load(fh)
        data = []
        for thm in theorems:
            head = {k: thm[k] for k in ("url", "commit", "file_path", "full_name", "start")}
            for i, tac in enumerate(thm["traced_tactics"]):
                ex = dict(head, tactic_idx=i,
                          context=Context(thm["file_path"], thm["full_name"], Pos(*thm["start"]), tac["state_before"]))
                all_pos = get_all_pos_premises(tac["annotated_tactic"], self.corpus)
                if not self.is_train:
                    data.append(dict(ex, all_pos_premises=all_pos))
                    continue
                for pos in all_pos:  # training: tactics without premises contribute nothing (datamodule.py:60-75)
                    data.append(dict(ex, pos_premise=pos, all_pos_premises=all_pos))
        return data

    def negative_pools(self, ex: Example) -> Tuple[List[int], List[int]]:
        """Premise indexes a training example may draw negatives from (datamodule.py:100-118): ``in_file`` = the other
        premises of the positive's file - when that is the theorem's own file only those that END BEFORE the theorem
        (strictly, unlike accessibility's <=); ``outside`` = the premises of every other (transitively) imported
        file, plus the theorem's own earlier premises when the positive lives elsewhere."""
        c = self.corpus
        ctx, pos = ex["context"], ex["pos_premise"]
        prem = c.all_premises
        f = c._index[ctx.path]
        own = [i for i in range(int(c._file_start[f]), int(c._file_start[f + 1]))
               if prem[i] != pos and prem[i].end < ctx.theorem_pos]
        same_file = pos.path == ctx.path
        in_file, outside = (own, []) if same_file else ([], own)
        for g in c._reach_ids(f):
            rng = range(int(c._file_start[g]), int(c._file_start[g + 1]))
            if c._files[g].path == pos.path:
                in_file += [i for i in rng if prem[i] != pos]
            else:
                outside += rng
        return in_file, outside

    def with_negatives(self, ex: Example) -> Example:
        """A copy of the example with ``neg_premises`` drawn as the reference draws them: up to
        ``num_in_file_negatives`` from the in-file pool, the rest from the outside pool (Python's ``random``, so
        ``random.seed`` governs it; ``ValueError`` from ``random.sample`` when a pool is too small, as upstream)."""
        in_file, outside = self.negative_pools(ex)
        k_in = min(len(in_file), self.num_in_file_negatives)
        picked = random.sample(in_file, k_in) + random.sample(outside, self.num_negatives - k_in)
        return dict(ex, neg_premises=[self.corpus.all_premises[i] for i in picked])

    def __len__(self) -> int:
        return len(self.data)

    def __getitem__(self, idx: int) -> Example:
        return self.with_negatives(self.data[idx]) if self.is_train else self.data[idx]

    def collate(self, examples: List[Example]) -> Batch:
        if self.is_train:
            return collate_train(examples, self.tokenizer, self.max_seq_len, self.num_negatives)
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

    def train_batches(self, batch_size: int, skip: int = 0) -> Iterator[Batch]:
        """One epoch as the reference's train DataLoader yields it (datamodule.py:255-264: shuffle=True,
        drop_last=True), negatives drawn per example as it is fetched.  ``skip``: the first ``skip`` batches (a resumed
        run's batches already trained on) are consumed at the INDEX level - their examples fetched, so the negative
        draws advance ``random`` exactly as an uninterrupted epoch does, but neither collated nor tokenised nor yielded."""
        order = list(range(len(self.data)))
        random.shuffle(order)
        for n, i in enumerate(range(0, len(order) - batch_size + 1, batch_size)):
            examples = [self[j] for j in order[i : i + batch_size]]
            if n >= skip:
                yield self.collate(examples)


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
    """``data_path`` holds ``{train,val,test}.json``; ``corpus_path`` is ``corpus.jsonl`` (datamodule.py:201-267)."""

    def __init__(self, data_path: str, corpus_path: str, eval_batch_size: int, max_seq_len: int, tokenizer,
                 corpus: Optional[Corpus] = None, num_negatives: int = 0, num_in_file_negatives: int = 0,
                 batch_size: int = 0, **_ignored) -> None:
        assert 0 <= num_in_file_negatives <= num_negatives
        self.data_path = data_path
        self.batch_size = batch_size
        self.eval_batch_size = eval_batch_size
        self.max_seq_len = max_seq_len
        self.num_negatives = num_negatives
        self.num_in_file_negatives = num_in_file_negatives
        self.tokenizer = tokenizer
        self.corpus = corpus if corpus is not None else Corpus(corpus_path)
        self.ds_train: Optional[RetrievalDataset] = None
        self.ds_val: Optional[RetrievalDataset] = None
        self.ds_pred: Optional[RetrievalDataset] = None

    def setup(self, stage: Optional[str] = None) -> None:
        def split(name):
            return os.path.join(self.data_path, f"{name}.json")

        if stage in (None, "fit"):
            self.ds_train = RetrievalDataset([split("train")], self.corpus, self.max_seq_len, self.tokenizer, is_train=True,
                                             num_negatives=self.num_negatives,
                                             num_in_file_negatives=self.num_in_file_negatives)
        if stage in (None, "fit", "validate"):
            self.ds_val = RetrievalDataset([split("val")], self.corpus, self.max_seq_len, self.tokenizer)
        if stage in (None, "fit", "predict"):
            self.ds_pred = RetrievalDataset([split(s) for s in ("train", "val", "test")], self.corpus,
                                            self.max_seq_len, self.tokenizer)

    def train_dataloader(self, skip: int = 0) -> Iterator[Batch]:
        return self.ds_train.train_batches(self.batch_size, skip)

    def val_dataloader(self) -> Iterator[Batch]:
        return self.ds_val.batches(self.eval_batch_size)

    def predict_dataloader(self) -> Iterator[Batch]:
        return self.ds_pred.batches(self.eval_batch_size)
