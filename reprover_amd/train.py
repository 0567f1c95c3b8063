"""The training step of the retriever on libreprover_hip.

Reference: ``retrieval/model.py:116-181`` (``forward`` / ``training_step`` / ``configure_optimizers``) differentiated by
autograd through ``_encode`` and HuggingFace's T5Stack, ``common.py:381-405`` (``get_optimizers``: ``torch.optim.AdamW(lr)``
under ``get_constant_schedule_with_warmup``; DeepSpeed's FusedAdam in adam_w_mode is the same update), and Lightning's
``gradient_clip_val`` (``retrieval/confs/cli_lean4_random.yaml:19``).  Here every piece is a HIP kernel behind the C ABI:

* ``HipT5Trainer`` owns the fp32 master parameters, the gradients and the AdamW moments as FLAT device tensors in the
  engine's canonical layout (``rp_train_param_layout``); ``named_parameters()`` / ``named_gradients()`` are views into them
  under the HuggingFace key names.
* one step = ``forward`` (all sequences of the batch packed as one varlen pass, activations saved in the workspace) →
  ``contrastive_mse`` + its backward on the [batch, D] embeddings → ``backward`` (encoder backward: every parameter's
  gradient) → ``optimizer_step`` (gradient norm, clipped AdamW over the flat buffers, bf16 compute copies refreshed).

Dropout (T5's ``dropout_rate``, 0.1 in the reference's training) is supported at HF's six sites with counter-based masks the
backward regenerates; ``dropout_rate=0`` (this class's default) is the deterministic step fixtures G11 / G12 pin, and the
dropout step is checked exactly against the oracle differentiated with the SAME masks.  Oracle: ``oracle/train_ref.py``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .encoder import _LAYER_KEYS, HipT5Encoder, _require_gpu

_LAYER_ORDER = ["ln_attn", "q", "k", "v", "o", "ln_ff", "wi_0", "wi_1", "wo"]  # rp_train.hip's per-layer order
REL_BIAS_KEY = "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"


def _rp_config(cfg: Dict) -> _lib.RpT5Config:
    return _lib.RpT5Config(
        cfg["vocab_size"], cfg["d_model"], cfg["d_kv"], cfg["num_heads"], cfg["d_ff"], cfg["num_layers"],
        cfg.get("relative_attention_num_buckets", 32), cfg.get("relative_attention_max_distance", 128),
        float(cfg.get("layer_norm_epsilon", 1e-6)))


def param_layout(cfg: Dict) -> List[Tuple[str, Tuple[int, ...], int]]:
    """[(HF key, shape, offset in the flat buffer)] in the engine's order + the total length (last entry, key "")."""
    lib = _lib.load()
    c = _rp_config(cfg)
    n = lib.rp_train_param_tensors(C.byref(c))
    off = (C.c_int64 * (n + 1))()
    _lib.check(lib.rp_train_param_layout(C.byref(c), off), "rp_train_param_layout")
    D, F, inner = cfg["d_model"], cfg["d_ff"], cfg["num_heads"] * cfg["d_kv"]
    shapes = {"ln_attn": (D,), "q": (inner, D), "k": (inner, D), "v": (inner, D), "o": (D, inner), "ln_ff": (D,),
              "wi_0": (F, D), "wi_1": (F, D), "wo": (D, F)}
    out = [("shared.weight", (cfg["vocab_size"], D), off[0]),
           (REL_BIAS_KEY, (cfg.get("relative_attention_num_buckets", 32), cfg["num_heads"]), off[1]),
           ("encoder.final_layer_norm.weight", (D,), off[2])]
    for i in range(cfg["num_layers"]):
        for j, fld in enumerate(_LAYER_ORDER):
            out.append((f"encoder.block.{i}.{_LAYER_KEYS[fld]}", shapes[fld], off[3 + 9 * i + j]))
    out.append(("", (), off[n]))
    return out


def warmup_factor(step_index: int, warmup_steps: int) -> float:
    """``get_constant_schedule_with_warmup``: multiplier of the ``step_index``-th optimizer step (0-based)."""
    return 1.0 if step_index >= warmup_steps else float(step_index) / float(max(1, warmup_steps))


def contrastive_mse_backward(context_emb: torch.Tensor, premise_embs: torch.Tensor, similarity: torch.Tensor,
                             label: torch.Tensor, d_ctx: Optional[torch.Tensor] = None,
                             d_prem: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(d loss / d context_emb [B, D], d loss / d premise_embs [P, D]) for loss = mse(context_emb @ premise_embs.T, label)."""
    lib = _lib.load()
    B, D = context_emb.shape
    P = premise_embs.shape[0]
    args = [t.to(torch.float32).contiguous() for t in (context_emb, premise_embs, similarity, label)]
    assert args[2].shape == (B, P) and args[3].shape == (B, P) and all(t.is_cuda for t in args)
    d_ctx = torch.empty_like(args[0]) if d_ctx is None else d_ctx
    d_prem = torch.empty_like(args[1]) if d_prem is None else d_prem
    with torch.cuda.device(context_emb.device):
        _lib.check(lib.rp_contrastive_mse_backward(*[_lib.ptr(t) for t in args], B, P, D, _lib.ptr(d_ctx), _lib.ptr(d_prem),
                                                   _lib.current_stream()), "rp_contrastive_mse_backward")
    return d_ctx, d_prem


def contrastive_mse(context_emb: torch.Tensor, premise_embs: torch.Tensor, label: torch.Tensor):
    """(loss 0-dim, similarity [B, P]) - retrieval/model.py:133-139 on fp32 device embeddings."""
    from .common import _workspace

    lib = _lib.load()
    B, D = context_emb.shape
    P = premise_embs.shape[0]
    dev = context_emb.device
    lab = label.to(device=dev, dtype=torch.float32).contiguous()
    assert lab.shape == (B, P), f"label {tuple(lab.shape)} != ({B}, {P})"
    loss = torch.empty((), dtype=torch.float32, device=dev)
    sim = torch.empty((B, P), dtype=torch.float32, device=dev)
    nbytes = lib.rp_contrastive_mse_workspace_bytes(B, P)
    ws = _workspace(dev, nbytes)
    with torch.cuda.device(dev):
        _lib.check(lib.rp_contrastive_mse(_lib.ptr(context_emb), _lib.ptr(premise_embs), _lib.ptr(lab), B, P, D,
                                          loss.data_ptr(), _lib.ptr(sim), _lib.ptr(ws), nbytes, _lib.current_stream()),
                   "rp_contrastive_mse")
    return loss, sim, lab


def pack_padded_groups(groups: Sequence[Tuple[torch.Tensor, torch.Tensor]]) -> Tuple[np.ndarray, np.ndarray]:
    """Right-padded (input_ids, attention_mask) batches, in order, → packed int32 ids + cu_seqlens of ALL their rows
    (the form ``rp_train_forward`` takes).  ``ValueError`` for a mask that is not right-padded or has an empty row, as
    the inference path raises."""
    rows, lens = [], []
    for ids, mask in groups:
        ids = np.asarray(ids.cpu().numpy() if isinstance(ids, torch.Tensor) else ids)
        mask = np.asarray(mask.cpu().numpy() if isinstance(mask, torch.Tensor) else mask) != 0
        n = mask.sum(1)
        if (n == 0).any() or (mask != (np.arange(mask.shape[1])[None, :] < n[:, None])).any():
            raise ValueError("attention_mask must be right-padded (1s then 0s) with at least one token per row, "
                             "as the tokenizer produces")
        for r in range(ids.shape[0]):
            rows.append(ids[r, : n[r]].astype(np.int32))
            lens.append(int(n[r]))
    cu = np.zeros(len(lens) + 1, dtype=np.int32)
    np.cumsum(lens, out=cu[1:])
    return np.concatenate(rows), cu


class HipT5Trainer:
    """fp32 masters + gradients + AdamW moments of a T5 encoder on one GPU, and the forward / backward / update
    launches over them."""

    def __init__(self, cfg: Dict, state_dict: Dict[str, torch.Tensor], device, lr: float = 0.0, warmup_steps: int = 0,
                 betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 gradient_clip_val: Optional[float] = None, out_dtype: torch.dtype = torch.bfloat16,
                 dropout_rate: float = 0.0, dropout_seed: int = 3407):
        if cfg.get("feed_forward_proj", "gated-gelu") != "gated-gelu":
            raise _lib.HipLibraryError(f"feed_forward_proj={cfg.get('feed_forward_proj')!r} is not implemented")
        self.device = _require_gpu(device)
        self.cfg = dict(cfg)
        self.lr, self.warmup_steps, self.betas, self.eps, self.weight_decay = lr, warmup_steps, betas, eps, weight_decay
        self.gradient_clip_val = gradient_clip_val
        # T5's dropout (HF config.dropout_rate; 0.1 in the reference's training): 0 = the deterministic step the fixtures
        # pin.  Masks are counter-based: forward number n of this trainer draws them from seed (dropout_seed, n), the
        # backward regenerates them.
        self.dropout_rate, self.dropout_seed = float(dropout_rate), int(dropout_seed)
        self._forwards = 0
        self.last_dropout_seed: Optional[int] = None
        self._lib = _lib.load()
        self.layout = param_layout(cfg)
        total = self.layout[-1][2]
        emb = "shared.weight" if "shared.weight" in state_dict else "encoder.embed_tokens.weight"
        with torch.cuda.device(self.device):
            self.params = torch.zeros(total, dtype=torch.float32, device=self.device)
            for key, shape, off in self.layout[:-1]:
                src = state_dict[emb if key == "shared.weight" else key].detach()
                assert tuple(src.shape) == tuple(shape), (key, tuple(src.shape), shape)
                self.params[off : off + src.numel()] = src.reshape(-1).to(device=self.device, dtype=torch.float32)
            self.grads = torch.zeros_like(self.params)
            self.exp_avg = torch.zeros_like(self.params)
            self.exp_avg_sq = torch.zeros_like(self.params)
            self.grad_norm = torch.zeros(1, dtype=torch.float32, device=self.device)
            self._norm_scratch = torch.empty(1024, dtype=torch.float32, device=self.device)
            handle = C.c_void_p()
            c = _rp_config(cfg)
            torch.cuda.synchronize(self.device)
            _lib.check(self._lib.rp_trainer_create(C.byref(c), _lib.ptr(self.params), C.byref(handle)), "rp_trainer_create")
        self._handle = handle
        self.steps = 0  # optimizer steps taken
        self._ws: Optional[torch.Tensor] = None
        self._pass = None  # (ids_d, cu_d, batch, T) of the last forward
        self.encoder = HipT5Encoder.from_handle(cfg, C.c_void_p(self._lib.rp_trainer_encoder(handle)), self.device,
                                                out_dtype, owner=self)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                self._lib.rp_trainer_destroy(h)
            except Exception:
                pass
            self._handle = None

    # -- views -------------------------------------------------------------------------------------------------
    def _views(self, flat: torch.Tensor) -> Iterator[Tuple[str, torch.Tensor]]:
        for key, shape, off in self.layout[:-1]:
            yield key, flat[off : off + int(np.prod(shape))].view(*shape)

    def named_parameters(self) -> Iterator[Tuple[str, torch.Tensor]]:
        """(HF key, fp32 view into the flat master buffer): writes go straight to the masters - call ``load_params()``
        afterwards so the bf16 compute copies follow."""
        return self._views(self.params)

    def named_gradients(self) -> Iterator[Tuple[str, torch.Tensor]]:
        return self._views(self.grads)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        sd = {k: v.detach().cpu().clone() for k, v in self.named_parameters()}
        sd["encoder.embed_tokens.weight"] = sd["shared.weight"]  # tied (HF:1074)
        return sd

    def save_training_state(self, path: str) -> None:
        """Everything a resumed run needs (what Lightning's ModelCheckpoint keeps for the reference): the fp32 masters,
        both AdamW moments and the step counter, as one safetensors file in the flat layout."""
        from safetensors.torch import save_file

        save_file({"params": self.params.cpu(), "exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu(),
                   "steps": torch.tensor([self.steps], dtype=torch.int64),
                   # the dropout stream: a resumed run continues the mask sequence instead of replaying it from forward 0
                   "dropout": torch.tensor([self.dropout_seed, self._forwards], dtype=torch.int64)}, path)

    def load_training_state(self, path: str) -> None:
        from safetensors.torch import load_file

        st = load_file(path)
        for key, mine in (("params", self.params), ("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
            if key not in st or st[key].numel() != mine.numel():
                raise ValueError(f"{path}: '{key}' has {st[key].numel() if key in st else 'no'} elements, this encoder's flat "
                                 f"layout has {mine.numel()} (the state belongs to another geometry)")
        if "dropout" in st:
            self.dropout_seed, self._forwards = int(st["dropout"][0]), int(st["dropout"][1])
        self.params.copy_(st["params"])
        self.exp_avg.copy_(st["exp_avg"])
        self.exp_avg_sq.copy_(st["exp_avg_sq"])
        self.steps = int(st["steps"][0])
        self.load_params()

    def load_params(self) -> None:
        with torch.cuda.device(self.device):
            _lib.check(self._lib.rp_trainer_load_params(self._handle, _lib.ptr(self.params), _lib.current_stream()),
                       "rp_trainer_load_params")

    # -- the step ------------------------------------------------------------------------------------------------
    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def forward(self, ids: np.ndarray, cu: np.ndarray) -> torch.Tensor:
        """Unit-norm fp32 embeddings [batch, d_model] of the packed sequences; the activations stay in the workspace
        for ``backward``."""
        batch, T = len(cu) - 1, int(cu[-1])
        assert batch > 0 and T > 0 and len(ids) == T and int(np.diff(cu).min()) > 0
        seed = (self.dropout_seed * 0x9E3779B1 + self._forwards * 0x85EBCA77 + 1) & 0xFFFFFFFF
        self._forwards += 1
        self.last_dropout_seed = seed if self.dropout_rate > 0 else None
        _lib.check(self._lib.rp_trainer_set_dropout(self._handle, self.dropout_rate, seed), "rp_trainer_set_dropout")
        with torch.cuda.device(self.device):
            ids_d = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int32)).to(self.device)
            cu_d = torch.from_numpy(np.ascontiguousarray(cu, dtype=np.int32)).to(self.device)
            nbytes = self._lib.rp_train_workspace_bytes(self._handle, T, batch)
            ws = self._workspace(nbytes)
            out = torch.empty((batch, self.cfg["d_model"]), dtype=torch.float32, device=self.device)
            _lib.check(self._lib.rp_train_forward(self._handle, _lib.ptr(ids_d), _lib.ptr(cu_d), batch, T, _lib.ptr(out),
                                                  _lib.ptr(ws), ws.numel(), _lib.current_stream()), "rp_train_forward")
        self._pass = (ids_d, cu_d, batch, T)
        return out

    def backward(self, d_emb: torch.Tensor) -> torch.Tensor:
        """d loss / d every parameter (the flat gradient buffer, overwritten) from d loss / d embeddings [batch, D]."""
        assert self._pass is not None, "backward() follows forward()"
        ids_d, cu_d, batch, T = self._pass
        assert d_emb.shape == (batch, self.cfg["d_model"]) and d_emb.dtype == torch.float32 and d_emb.is_contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.rp_train_backward(self._handle, _lib.ptr(self.params), _lib.ptr(ids_d), _lib.ptr(cu_d), batch,
                                                   T, _lib.ptr(d_emb), _lib.ptr(self.grads), _lib.ptr(self._ws),
                                                   self._ws.numel(), _lib.current_stream()), "rp_train_backward")
        return self.grads

    def current_lr(self) -> float:
        return self.lr * warmup_factor(self.steps, self.warmup_steps)

    def optimizer_step(self) -> None:
        """clip_grad_norm_ (when ``gradient_clip_val`` is set) + one AdamW update of every parameter + refresh of the
        compute copies; the schedule's factor is the one in force for this step (0 for the first step of a warm-up)."""
        lr = self.current_lr()
        self.steps += 1
        n = self.params.numel()
        with torch.cuda.device(self.device):
            s = _lib.current_stream()
            clip = self.gradient_clip_val is not None and self.gradient_clip_val > 0
            _lib.check(self._lib.rp_grad_norm(_lib.ptr(self.grads), n, _lib.ptr(self.grad_norm), _lib.ptr(self._norm_scratch), s),
                       "rp_grad_norm")
            _lib.check(self._lib.rp_adamw_step_clipped(
                _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), n, self.steps,
                lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                _lib.ptr(self.grad_norm) if clip else None, float(self.gradient_clip_val or 0.0), s), "rp_adamw_step_clipped")
        self.load_params()

    def contrastive_step(self, groups: Sequence[Tuple[torch.Tensor, torch.Tensor]], label: torch.Tensor):
        """forward + loss + backward of the reference's ``forward`` (model.py:116-140) for ``groups`` = [(context_ids,
        context_mask), (pos_ids, pos_mask), (neg_ids_0, neg_mask_0), ...]: returns (loss, similarity); the gradients are
        in ``self.grads``."""
        ids, cu = pack_padded_groups(groups)
        emb = self.forward(ids, cu)
        B = groups[0][0].shape[0]
        ctx, prem = emb[:B], emb[B:]
        loss, sim, lab = contrastive_mse(ctx, prem, label)
        d_emb = torch.empty_like(emb)
        contrastive_mse_backward(ctx, prem, sim, lab, d_emb[:B], d_emb[B:])
        self.backward(d_emb)
        return loss, sim


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
