"""Oracle (test infrastructure only): evaluation-side restatements.

* ``load_eval_examples``  — retrieval/datamodule.py:44-90 with ``is_train=False`` + common.py:341-354
* ``eval_predictions``    — retrieval/evaluate.py:13-43 (R@1, R@10, MRR)
* ``validation_metrics``  — retrieval/model.py:227-268 (Recall@k for every k, MRR)
Premises are identified by their index in ``CorpusRef.all_premises``.
"""
from __future__ import annotations

import json
from typing import Dict, List, Sequence, Tuple

import numpy as np

from .common_ref import CorpusRef, Pos


def all_pos_premise_indexes(annot_tac, corpus: CorpusRef, where: Dict[int, int]) -> List[int]:
    """common.py:341-354: provenances → located premises (deduplicated), as sorted indexes."""
    _, provenances = annot_tac
    found = set()
    for prov in provenances:
        p = corpus.locate_premise(prov["def_path"], Pos(*prov["def_pos"]))
        if p is not None:
            found.add(where[id(p)])
    return sorted(found)


def load_eval_examples(data_path: str, corpus: CorpusRef) -> List[dict]:
    where = {id(p): i for i, p in enumerate(corpus.all_premises)}
    out = []
    for thm in json.load(open(data_path)):
        for i, tac in enumerate(thm["traced_tactics"]):
            out.append({
                "file_path": thm["file_path"], "full_name": thm["full_name"], "start": thm["start"],
                "tactic_idx": i, "state": tac["state_before"],
                "all_pos_premises": all_pos_premise_indexes(tac["annotated_tactic"], corpus, where),
            })
    return out


def eval_predictions(examples: Sequence[dict], retrieved: Sequence[Sequence[int]]) -> Tuple[float, float, float]:
    r1, r10, mrr = [], [], []
    for ex, got in zip(examples, retrieved):
        pos = set(ex["all_pos_premises"])
        if not pos:
            continue  # evaluate.py:24-25
        r1.append(float(got[0] in pos) / len(pos))
        r10.append(len(pos & set(got[:10])) / len(pos))
        rr = 0.0
        for j, p in enumerate(got):
            if p in pos:
                rr = 1.0 / (j + 1)
                break
        mrr.append(rr)
    return 100 * float(np.mean(r1)), 100 * float(np.mean(r10)), float(np.mean(mrr))


def validation_metrics(pos_batch: Sequence[Sequence[int]], retrieved: Sequence[Sequence[int]], k: int):
    recall = [[] for _ in range(k)]
    mrr = []
    for pos, got in zip(pos_batch, retrieved):
        pos = set(pos)
        if not pos:
            continue  # model.py:237-238
        first = False
        for j in range(k):
            recall[j].append(len(pos & set(got[: j + 1])) / len(pos))  # model.py:244-245
            if got[j] in pos and not first:
                mrr.append(1.0 / (j + 1))
                first = True
        if not first:
            mrr.append(0.0)
    return [100 * float(np.mean(r)) for r in recall], float(np.mean(mrr))
