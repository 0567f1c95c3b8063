"""CPU restatement of the training step of the retriever (test infrastructure only - see oracle/__init__.py).

Follows lean-dojo/ReProver:
  * retrieval/model.py:116-140  ``forward``: similarity = context_emb @ cat(pos_emb, *neg_embs).T,
    loss = F.mse_loss(similarity, label);
  * retrieval/datamodule.py:160-175  the label matrix of ``collate`` (is_train=True);
  * retrieval/model.py:142-160  ``training_step`` = that loss, differentiated by autograd;
  * common.py:381-405  ``get_optimizers``: ``torch.optim.AdamW(parameters, lr=lr)`` (defaults betas (0.9, 0.999), eps 1e-8,
    weight_decay 1e-2) under transformers' ``get_constant_schedule_with_warmup``.
Pinned by tests/golden/g10_train_forward.npz (the reference's own collate + forward, HuggingFace fp32) and
tests/golden/g11_train_backward.npz (the reference's loss.backward() and two AdamW steps; dropout off).
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


# ------------------------------------------------------------------------------------------------------------------
# backward + optimizer: what a HIP training step will be checked against
# ------------------------------------------------------------------------------------------------------------------
TIED = "encoder.embed_tokens.weight"  # HF ties it to shared.weight: one parameter, one gradient


def forward_backward(cfg: Dict, sd: Dict[str, torch.Tensor], context_texts: List[str], pos_texts: List[str],
                     neg_texts: List[List[str]], label: np.ndarray, max_seq_len: int, drop_for_group=None):
    """(loss, {parameter name: d loss / d parameter}) in torch fp32 - autograd over the same restatement the forward
    oracle evaluates (t5_ref._encode_texts).  Dropout is the identity (the reference trains with T5's dropout 0.1,
    which is stochastic and therefore not something a golden vector can pin) unless ``drop_for_group`` is given:
    ``drop_for_group(g)`` returns the mask object (see t5_ref._encoder_forward) of encode g = 0 contexts, 1 positives,
    2.. the negative lists - the training step with GIVEN dropout masks, differentiated exactly."""
    W = {k: v.detach().clone().float().requires_grad_(True) for k, v in sd.items() if k != TIED}
    with torch.enable_grad():
        dg = drop_for_group or (lambda g: None)
        ctx = t5_ref._encode_texts(cfg, W, context_texts, max_seq_len, len(context_texts), dg(0))
        prem = [t5_ref._encode_texts(cfg, W, pos_texts, max_seq_len, len(pos_texts), dg(1))]
        for g, texts in enumerate(neg_texts):
            prem.append(t5_ref._encode_texts(cfg, W, texts, max_seq_len, len(texts), dg(2 + g)))
        sim = ctx @ torch.cat(prem, dim=0).T
        loss = torch.nn.functional.mse_loss(sim, torch.from_numpy(np.asarray(label, dtype=np.float32)))
    loss.backward()
    grads = {k: (w.grad if w.grad is not None else torch.zeros_like(w)).detach().numpy() for k, w in W.items()}
    return float(loss.detach()), grads


def warmup_factor(step_index: int, warmup_steps: int) -> float:
    """transformers.get_constant_schedule_with_warmup: the multiplier in force for the ``step_index``-th optimizer step
    (0-based: LambdaLR starts at lambda(0), so with warmup > 0 the very first step has learning rate 0)."""
    return 1.0 if step_index >= warmup_steps else float(step_index) / float(max(1, warmup_steps))


def adamw_step(param: np.ndarray, grad: np.ndarray, m: np.ndarray, v: np.ndarray, t: int, lr: float,
               betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
    """One torch.optim.AdamW update (decoupled weight decay, bias-corrected moments), t = 1, 2, ...; returns the new
    (param, m, v).  float64 inside so that the restatement is not the source of a mismatch."""
    p, g = param.astype(np.float64), grad.astype(np.float64)
    p = p * (1.0 - lr * weight_decay)
    m = betas[0] * m.astype(np.float64) + (1.0 - betas[0]) * g
    v = betas[1] * v.astype(np.float64) + (1.0 - betas[1]) * g * g
    m_hat = m / (1.0 - betas[0] ** t)
    denom = np.sqrt(v) / np.sqrt(1.0 - betas[1] ** t) + eps
    p = p - lr * m_hat / denom
    return p.astype(np.float32), m, v
