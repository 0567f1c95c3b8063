"""The two ends of the training step around the encoder (the encoder's own backward is not built):
the backward of the contrastive-MSE loss and the AdamW update, as HIP kernels behind the C ABI.

Reference: ``retrieval/model.py:116-140`` (loss), ``common.py:381-405`` (``get_optimizers``: ``torch.optim.AdamW(lr)`` under
``get_constant_schedule_with_warmup``).  Oracle: ``oracle/train_ref.py``, fixture G11."""
from typing import Tuple

import torch

from . import _lib


def contrastive_mse_backward(context_emb: torch.Tensor, premise_embs: torch.Tensor, similarity: torch.Tensor,
                             label: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(d loss / d context_emb [B, D], d loss / d premise_embs [P, D]) for loss = mse(context_emb @ premise_embs.T, label)."""
    lib = _lib.load()
    B, D = context_emb.shape
    P = premise_embs.shape[0]
    args = [t.to(torch.float32).contiguous() for t in (context_emb, premise_embs, similarity, label)]
    assert args[2].shape == (B, P) and args[3].shape == (B, P) and all(t.is_cuda for t in args)
    d_ctx, d_prem = torch.empty_like(args[0]), torch.empty_like(args[1])
    with torch.cuda.device(context_emb.device):
        _lib.check(lib.rp_contrastive_mse_backward(*[_lib.ptr(t) for t in args], B, P, D, _lib.ptr(d_ctx), _lib.ptr(d_prem),
                                                   _lib.current_stream()), "rp_contrastive_mse_backward")
    return d_ctx, d_prem


def warmup_factor(step_index: int, warmup_steps: int) -> float:
    """``get_constant_schedule_with_warmup``: multiplier of the ``step_index``-th optimizer step (0-based)."""
    return 1.0 if step_index >= warmup_steps else float(step_index) / float(max(1, warmup_steps))


class AdamW:
    """``torch.optim.AdamW(params, lr)`` (the reference's optimizer outside DeepSpeed, common.py:395) with the constant
    schedule after a linear warm-up (common.py:397), over flat fp32 device tensors; the update is one HIP kernel per
    tensor.  Gradients are supplied by the caller (``step(grads)``)."""

    def __init__(self, params, lr: float, warmup_steps: int = 0, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2):
        self.params = list(params)
        for p in self.params:
            assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
        self.lr, self.warmup_steps, self.betas, self.eps, self.weight_decay = lr, warmup_steps, betas, eps, weight_decay
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.steps = 0

    def step(self, grads) -> None:
        lib = _lib.load()
        lr = self.lr * warmup_factor(self.steps, self.warmup_steps)
        self.steps += 1
        for p, g, m, v in zip(self.params, grads, self.exp_avg, self.exp_avg_sq):
            assert g.shape == p.shape and g.dtype == torch.float32 and g.is_contiguous()
            with torch.cuda.device(p.device):
                _lib.check(lib.rp_adamw_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), p.numel(), self.steps, lr,
                                             self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                             _lib.current_stream()), "rp_adamw_step")
