"""CPU restatement of the training FORWARD of the retriever (test infrastructure only - see oracle/__init__.py).

Follows lean-dojo/ReProver:
  * retrieval/model.py:116-140  ``forward``: similarity = context_emb @ cat(pos_emb, *neg_embs).T,
    loss = F.mse_loss(similarity, label);
  * retrieval/datamodule.py:160-175  the label matrix of ``collate`` (is_train=True).
Pinned by tests/golden/g10_train_forward.npz (the reference's own collate + forward, HuggingFace fp32).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

from . import t5_ref


def label_matrix(pos_keys: Sequence, neg_keys: Sequence[Sequence], all_pos_keys: Sequence[Sequence]) -> np.ndarray:
    """pos_keys[j]: example j's positive premise; neg_keys[j][i]: its i-th negative; all_pos_keys[j]: every
    positive premise of example j (any hashable identity).  datamodule.py:160-175."""
    n = len(pos_keys)
    nneg = len(neg_keys[0]) if n else 0
    label = np.zeros((n, n * (1 + nneg)), dtype=np.float32)
    for j in range(n):
        for k in range(n * (1 + nneg)):
            prem = pos_keys[k] if k < n else neg_keys[k % n][k // n - 1]
            label[j, k] = float(prem in all_pos_keys[j])
    return label


def forward_loss(cfg: Dict, sd: Dict[str, torch.Tensor], context_texts: List[str], pos_texts: List[str],
                 neg_texts: List[List[str]], label: np.ndarray, max_seq_len: int):
    """(loss, similarity [B, P]) in torch fp32; neg_texts[i] = the i-th negative of every example."""
    ctx = t5_ref.encode_texts(cfg, sd, context_texts, max_seq_len, len(context_texts))
    prem = [t5_ref.encode_texts(cfg, sd, pos_texts, max_seq_len, len(pos_texts))]
    for texts in neg_texts:
        prem.append(t5_ref.encode_texts(cfg, sd, texts, max_seq_len, len(texts)))
    sim = ctx @ torch.cat(prem, dim=0).T
    loss = torch.nn.functional.mse_loss(sim, torch.from_numpy(np.asarray(label, dtype=np.float32)))
    return float(loss), sim.numpy()
