"""R@1 / R@10 / MRR of a ``predictions.pickle`` — the reference's ``retrieval/evaluate.py``
(lean-dojo/ReProver retrieval/evaluate.py:13-43), same CLI (``--preds-file --data-path``)."""
from __future__ import annotations

import argparse
import json
import os
import pickle
from typing import Dict, List, Sequence, Tuple

import numpy as np


def _eval(data, preds_map) -> Tuple[float, float, float]:
    """Per traced tactic with at least one ground-truth premise: R@1 and R@10 are hits / #positives
    (in %), MRR the reciprocal rank of the first hit (0 when none is retrieved)."""
    r1, r10, mrr = [], [], []
    for thm in data:
        for i, _ in enumerate(thm["traced_tactics"]):
            pred = preds_map[(thm["file_path"], thm["full_name"], tuple(thm["start"]), i)]
            pos = set(pred["all_pos_premises"])
            if not pos:
                continue
            got = pred["retrieved_premises"]
            r1.append(float(got[0] in pos) / len(pos))
            r10.append(float(len(pos.intersection(got[:10]))) / len(pos))
            rank = next((j for j, p in enumerate(got) if p in pos), None)
            mrr.append(0.0 if rank is None else 1.0 / (rank + 1))
    return 100 * float(np.mean(r1)), 100 * float(np.mean(r10)), float(np.mean(mrr))


def recall_and_mrr(all_pos_premises_batch: Sequence[Sequence], retrieved_batch: Sequence[Sequence], num_retrieved: int):
    """What ``validation_step`` logs (retrieval/model.py:227-268): Recall@1..num_retrieved (in %) and
    MRR over the examples that have ground-truth premises; also returns how many those are."""
    recall: List[List[float]] = [[] for _ in range(num_retrieved)]
    mrr: List[float] = []
    n = 0
    for pos, got in zip(all_pos_premises_batch, retrieved_batch):
        pos = set(pos)
        if not pos:
            continue
        n += 1
        first = None
        hits = 0
        for j in range(num_retrieved):
            if got[j] in pos:
                hits += 1  # retrieved lists hold distinct premises, so this is |pos ∩ got[:j+1]|
                if first is None:
                    first = j
            recall[j].append(float(hits) / len(pos))
        mrr.append(0.0 if first is None else 1.0 / (first + 1))
    return [100 * float(np.mean(r)) for r in recall], float(np.mean(mrr)), n


def load_preds_map(preds_file: str) -> Dict:
    with open(preds_file, "rb") as fh:
        preds = pickle.load(fh)
    preds_map = {(p["file_path"], p["full_name"], tuple(p["start"]), p["tactic_idx"]): p for p in preds}
    assert len(preds) == len(preds_map), "Duplicate predictions found!"
    return preds_map


def main(argv=None) -> None:
    parser = argparse.ArgumentParser(description="Script for evaluating the premise retriever.")
    parser.add_argument("--preds-file", type=str, required=True, help="Path to the retriever's predictions file.")
    parser.add_argument("--data-path", type=str, required=True,
                        help="Path to the directory containing the train/val/test splits.")
    args = parser.parse_args(argv)
    preds_map = load_preds_map(args.preds_file)
    for split in ("train", "val", "test"):
        data_path = os.path.join(args.data_path, f"{split}.json")
        with open(data_path) as fh:
            data = json.load(fh)
        r1, r10, mrr = _eval(data, preds_map)
        print(f"{data_path}: R@1 = {r1} %, R@10 = {r10} %, MRR = {mrr}")


if __name__ == "__main__":
    main()
