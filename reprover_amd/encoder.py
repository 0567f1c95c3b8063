"""HipT5Encoder: the ByT5/T5 text encoder of the retrieval path, running on libreprover_hip.

Stands in for ``AutoModelForTextEncoding.from_pretrained(...)`` → ``T5EncoderModel``
(retrieval/model.py:45) *together with* the masked mean-pool + L2 normalise of
``PremiseRetriever._encode`` (model.py:92-114): one call = ``rp_encode_varlen``.
PyTorch tensors are containers only (device memory + stream).
"""
from __future__ import annotations

import ctypes as C
import json
import os
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib

_LAYER_KEYS = {
    "ln_attn": "layer.0.layer_norm.weight",
    "q": "layer.0.SelfAttention.q.weight",
    "k": "layer.0.SelfAttention.k.weight",
    "v": "layer.0.SelfAttention.v.weight",
    "o": "layer.0.SelfAttention.o.weight",
    "ln_ff": "layer.1.layer_norm.weight",
    "wi_0": "layer.1.DenseReluDense.wi_0.weight",
    "wi_1": "layer.1.DenseReluDense.wi_1.weight",
    "wo": "layer.1.DenseReluDense.wo.weight",
}


def _require_gpu(device) -> torch.device:
    device = torch.device(device)
    if device.type == "cuda" and device.index is None and torch.cuda.is_available():
        device = torch.device("cuda", torch.cuda.current_device())  # tensors report an explicit index
    if device.type != "cuda" or not torch.cuda.is_available():
        raise _lib.HipLibraryError(
            "the MI355X retrieval engine needs a HIP device; there is no CPU path "
            f"(requested device={device}, torch.cuda.is_available()={torch.cuda.is_available()})"
        )
    return device


class HipT5Encoder:
    """T5 encoder weights resident on one GPU + the packed varlen forward."""

    def __init__(self, cfg: Dict, state_dict: Dict[str, torch.Tensor], device, dtype: torch.dtype = torch.bfloat16,
                 max_tokens_per_pass: int = 1 << 18, keep_master_weights: bool = True):
        if cfg.get("feed_forward_proj", "gated-gelu") != "gated-gelu":
            raise _lib.HipLibraryError(f"feed_forward_proj={cfg.get('feed_forward_proj')!r} is not implemented")
        self.device = _require_gpu(device)
        assert dtype in (torch.bfloat16, torch.float32)
        self.dtype = dtype  # dtype of the embeddings handed back (compute is bf16 MFMA / fp32 accumulate)
        self.cfg = dict(cfg)
        self.config = SimpleNamespace(hidden_size=cfg["d_model"], **cfg)
        self.max_tokens_per_pass = int(max_tokens_per_pass)
        lib = _lib.load()
        c = _lib.RpT5Config(
            cfg["vocab_size"], cfg["d_model"], cfg["d_kv"], cfg["num_heads"], cfg["d_ff"], cfg["num_layers"],
            cfg.get("relative_attention_num_buckets", 32), cfg.get("relative_attention_max_distance", 128),
            float(cfg.get("layer_norm_epsilon", 1e-6)),
        )
        with torch.cuda.device(self.device):
            keep = []

            def dev(name: str) -> int:
                t = state_dict[name].detach().to(device=self.device, dtype=torch.float32).contiguous()
                keep.append(t)
                return t.data_ptr()

            emb = "shared.weight" if "shared.weight" in state_dict else "encoder.embed_tokens.weight"
            layers = (_lib.RpT5LayerWeights * cfg["num_layers"])()
            for i in range(cfg["num_layers"]):
                for fld, key in _LAYER_KEYS.items():
                    setattr(layers[i], fld, dev(f"encoder.block.{i}.{key}"))
            w = _lib.RpT5Weights(
                dev(emb),
                dev("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"),
                dev("encoder.final_layer_norm.weight"),
                layers,
            )
            handle = C.c_void_p()
            torch.cuda.synchronize(self.device)
            _lib.check(lib.rp_encoder_create(C.byref(c), C.byref(w), _lib.RP_DT_F32, C.byref(handle)),
                       "rp_encoder_create")
            del keep
        self._handle = handle
        self._lib = lib
        self._owner = None
        self._ws: Optional[torch.Tensor] = None
        self._pending_meta: list = []
        # The fp32 weights a training step (reprover_amd/train.py::HipT5Trainer) or ``save_pretrained`` starts from.  Kept on
        # the HOST (a dict of device tensors would pin a second, fp32 copy of the weights in HBM beside the packed bf16
        # one for the encoder's lifetime); ``keep_master_weights=False`` keeps nothing - what a process that only
        # retrieves wants (the reference runs many retriever actors per node), and ``train_engine`` then reloads from
        # ``self.source_path`` when the encoder came from a checkpoint directory.
        self.source_path: Optional[str] = None
        if keep_master_weights:
            self._state_dict_cpu = {k: (v.detach().to("cpu") if v.is_cuda else v) for k, v in state_dict.items()}
        else:
            self._state_dict_cpu = None

    @classmethod
    def from_handle(cls, cfg: Dict, handle, device, dtype: torch.dtype, owner) -> "HipT5Encoder":
        """The inference engine over an RpEncoder owned by someone else (``owner``: a ``HipT5Trainer``, kept alive)."""
        self = cls.__new__(cls)
        self.device = _require_gpu(device)
        self.dtype = dtype
        self.cfg = dict(cfg)
        self.config = SimpleNamespace(hidden_size=cfg["d_model"], **cfg)
        self.max_tokens_per_pass = 1 << 18
        self._state_dict_cpu = None
        self.source_path = None
        self._lib = _lib.load()
        self._handle = handle
        self._owner = owner
        self._ws = None
        self._pending_meta = []
        return self

    # -- construction helpers ---------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path: str, device, dtype: torch.dtype = torch.bfloat16,
                        keep_master_weights: bool = True) -> "HipT5Encoder":
        """Load a HuggingFace T5/ByT5 checkpoint directory (config.json + model.safetensors or
        pytorch_model.bin; key names SURVEY.md App. B.5; decoder.* keys are ignored)."""
        if not os.path.isdir(path):
            raise FileExistsError(f"Checkpoint {path} does not exist.")  # common.py:409-410's convention
        with open(os.path.join(path, "config.json")) as fh:
            hf = json.load(fh)
        cfg = dict(
            vocab_size=hf["vocab_size"], d_model=hf["d_model"], d_kv=hf["d_kv"], num_heads=hf["num_heads"],
            d_ff=hf["d_ff"], num_layers=hf["num_layers"],
            relative_attention_num_buckets=hf.get("relative_attention_num_buckets", 32),
            relative_attention_max_distance=hf.get("relative_attention_max_distance", 128),
            layer_norm_epsilon=hf.get("layer_norm_epsilon", 1e-6),
            feed_forward_proj=hf.get("feed_forward_proj", "relu"),
            dropout_rate=hf.get("dropout_rate", 0.1),
        )
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file

            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        sd = {k: v for k, v in sd.items() if not k.startswith("decoder.") and not k.startswith("lm_head")}
        enc = cls(cfg, sd, device, dtype, keep_master_weights=keep_master_weights)
        enc.source_path = path
        return enc

    def master_weights(self) -> Optional[Dict[str, torch.Tensor]]:
        """The fp32 weights this engine was built from: the kept host dict, or re-read from ``source_path``."""
        if self._state_dict_cpu is None and self.source_path is not None:
            st = os.path.join(self.source_path, "model.safetensors")
            if os.path.exists(st):
                from safetensors.torch import load_file

                sd = load_file(st)
            else:
                sd = torch.load(os.path.join(self.source_path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
            self._state_dict_cpu = {k: v for k, v in sd.items() if not k.startswith("decoder.") and not k.startswith("lm_head")}
        return self._state_dict_cpu

    def save_pretrained(self, path: str) -> None:
        """Write an HF-layout checkpoint directory (config.json + model.safetensors) that ``from_pretrained`` - and
        ``transformers``' T5EncoderModel - read back: the trainer's current fp32 masters when this engine belongs to a
        ``HipT5Trainer``, else the weights it was built from."""
        from safetensors.torch import save_file

        sd = self._owner.state_dict() if self._owner is not None else self.master_weights()
        if sd is None:
            raise RuntimeError("this encoder holds no fp32 weights to save")
        os.makedirs(path, exist_ok=True)
        hf = {k: v for k, v in self.cfg.items()}
        hf.update(model_type="t5", architectures=["T5EncoderModel"], dropout_rate=self.cfg.get("dropout_rate", 0.1),
                  tie_word_embeddings=False, num_decoder_layers=0, is_encoder_decoder=False)
        with open(os.path.join(path, "config.json"), "w") as fh:
            json.dump(hf, fh, indent=1)
        keep = {k: v.detach().to(torch.float32).cpu().contiguous().clone() for k, v in sd.items()
                if k != "encoder.embed_tokens.weight"}  # tied to shared.weight: stored once
        save_file(keep, os.path.join(path, "model.safetensors"))

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value and getattr(self, "_owner", None) is None:
            try:
                self._lib.rp_encoder_destroy(h)
            except Exception:
                pass
            self._handle = None

    # -- forward ----------------------------------------------------------------------------------
    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def encode_packed_device(self, ids: torch.Tensor, cu: torch.Tensor, batch: int, total: int, max_len: int,
                             out: torch.Tensor) -> None:
        """One ``rp_encode_varlen`` launch sequence; everything already on the device.
        ``out`` = [batch, d_model] rows (may be a slice of a larger matrix), dtype f32 or bf16."""
        assert ids.dtype == torch.int32 and cu.dtype == torch.int32
        assert out.is_contiguous() and out.shape == (batch, self.cfg["d_model"])
        nbytes = self._lib.rp_encoder_workspace_bytes(self._handle, total, batch)
        ws = self._workspace(nbytes)
        out_dt = _lib.RP_DT_BF16 if out.dtype == torch.bfloat16 else _lib.RP_DT_F32
        with torch.cuda.device(self.device):
            _lib.check(
                self._lib.rp_encode_varlen(self._handle, _lib.ptr(ids), _lib.ptr(cu), batch, total, max_len,
                                           out.data_ptr(), out_dt, _lib.ptr(ws), ws.numel(),
                                           _lib.current_stream()),
                "rp_encode_varlen",
            )

    def encode_packed(self, ids: np.ndarray, cu: np.ndarray, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Encode sequences given as packed host ids (int32 [T]) + cu_seqlens (int32 [B+1]);
        splits into passes of at most ``max_tokens_per_pass`` tokens.  Returns [B, d_model]."""
        B = len(cu) - 1
        D = self.cfg["d_model"]
        if out is None:
            out = torch.empty((B, D), dtype=self.dtype, device=self.device)
        lens = np.diff(cu)
        assert B > 0 and lens.min() > 0, "every sequence needs at least the EOS token"
        b0 = 0
        while b0 < B:
            b1 = int(np.searchsorted(cu, cu[b0] + self.max_tokens_per_pass, side="right")) - 1
            b1 = min(max(b1, b0 + 1), B)
            t0, t1 = int(cu[b0]), int(cu[b1])
            ids_d, cu_d = self._stage(np.asarray(ids[t0:t1], dtype=np.int32), (cu[b0 : b1 + 1] - cu[b0]).astype(np.int32))
            self.encode_packed_device(ids_d, cu_d, b1 - b0, t1 - t0, int(lens[b0:b1].max()), out[b0:b1])
            b0 = b1
        return out

    def encode_padded(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, defer_check: bool = False,
                      out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """Drop-in for ``_encode(input_ids, attention_mask)`` with right-padded [B, L] inputs.  DEVICE tensors: one
        ``rp_encode_padded`` launch sequence — lengths, cu_seqlens, id compaction and the right-padding check all
        happen on the device; no torch kernels, no host round trip.  HOST tensors are packed on the host and take the
        packed entry point (``_encode_host_batch``: the right-padding check raises at once).  On the device form the
        check's verdict arrives asynchronously:
        by default it is read back here (one 16-byte copy) and ``ValueError`` is raised, as the packed path does,
        for a mask that is not right-padded or an empty row; with ``defer_check=True`` the call is launch-only and
        the caller runs ``raise_pending()`` at its next synchronisation point (``predict_step`` does)."""
        B, L = input_ids.shape
        assert attention_mask.shape == (B, L)
        if not input_ids.is_cuda and not attention_mask.is_cuda:
            # Host batches (the collate's output) are packed on the host - lengths from the mask, right-padding check,
            # ids of the real tokens only - and go through the packed entry point: ~4 bytes per REAL token cross PCIe
            # through pinned staging memory, asynchronously (a `.to(device)` of the padded int64 pair from pageable
            # memory moved 16 bytes per padded position and blocked the host until the stream reached the copy, i.e.
            # until the previous batch's encode and search had finished).  Same kernels, same bits as the device form.
            return self._encode_host_batch(input_ids, attention_mask, out_dtype)
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        mask = attention_mask.to(device=self.device, dtype=torch.int64).contiguous()
        if B * L > self.max_tokens_per_pass and B > 1:  # rare (huge padded batches): chunk the batch dimension;
            step = max(1, self.max_tokens_per_pass // L)  # a single row always runs as one pass
            return torch.cat([self.encode_padded(ids[i : i + step], mask[i : i + step], defer_check, out_dtype)
                              for i in range(0, B, step)])
        out = torch.empty((B, self.cfg["d_model"]), dtype=out_dtype or self.dtype, device=self.device)
        meta = torch.empty(4, dtype=torch.int32, device=self.device)
        self.encode_padded_into(ids, mask, out, meta)
        self._pending_meta.append(meta)
        if not defer_check:
            self.raise_pending()
        return out

    _STAGING_SLOTS = 4

    def _stage(self, *arrays: np.ndarray):
        """Device views of int32 host arrays uploaded by ONE asynchronous copy from a ring of pinned staging buffers (a
        slot is reused four uploads later, after its copy's event; page-locking is slow, so slots are sized generously
        once)."""
        sizes = [int(a.size) for a in arrays]
        total = sum(sizes)
        ring = self.__dict__.setdefault("_staging", [])
        nxt = self.__dict__.get("_staging_next", 0)
        cap = max(1 << 18, 1 << max(total - 1, 1).bit_length())
        if len(ring) < self._STAGING_SLOTS:
            ring.append([torch.empty(cap, dtype=torch.int32).pin_memory(), None])
            slot = ring[-1]
        else:
            slot = ring[nxt]
            self._staging_next = (nxt + 1) % self._STAGING_SLOTS
            if slot[1] is not None:
                slot[1].synchronize()
            if slot[0].numel() < total:
                slot[0] = torch.empty(cap, dtype=torch.int32).pin_memory()
        h = slot[0].numpy()
        off = 0
        for a, n in zip(arrays, sizes):
            h[off : off + n] = a.reshape(-1)
            off += n
        with torch.cuda.device(self.device):
            d = torch.empty(total, dtype=torch.int32, device=self.device)
            d.copy_(slot[0][:total], non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record(torch.cuda.current_stream(self.device))
        out, off = [], 0
        for n in sizes:
            out.append(d[off : off + n])
            off += n
        return out

    @staticmethod
    def _pack_host(input_ids: torch.Tensor, attention_mask: torch.Tensor):
        """(packed int32 ids of the real tokens, per-row lengths) of one right-padded host batch; ``ValueError`` for a mask
        that is not right-padded or has an empty row (what the device form reports through its verdict word)."""
        ids = input_ids.numpy()
        mask = attention_mask.numpy() != 0
        L = ids.shape[1]
        lens = mask.sum(1)
        if (lens == 0).any() or (mask != (np.arange(L)[None, :] < lens[:, None])).any():
            raise ValueError("attention_mask must be right-padded (1s then 0s) with at least one token per row, "
                             "as the tokenizer produces")
        return ids[mask].astype(np.int32), lens  # row-major: sequence after sequence

    def _encode_host_batch(self, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                           out_dtype: Optional[torch.dtype]) -> torch.Tensor:
        return self.encode_padded_many([(input_ids, attention_mask)], out_dtype)

    def encode_padded_many(self, batches, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """Several right-padded HOST batches ``[(input_ids [B_i, L_i], attention_mask [B_i, L_i]), ...]`` as ONE packed
        pass (rows in order: the embeddings of batch 0, then batch 1, ...).  What ``predict_step`` uses to run the
        reference's 64-state eval batches at the pass size the GPU is efficient at: four 64-state passes cost 10 % more
        GPU time than one 256-state pass (tools/pass_size_profile.py), and a row's embedding does not depend on what
        else is in the pass."""
        parts = [self._pack_host(i, m) for i, m in batches]
        packed = np.concatenate([p for p, _ in parts]) if len(parts) > 1 else parts[0][0]
        lens = np.concatenate([n for _, n in parts]) if len(parts) > 1 else parts[0][1]
        B = len(lens)
        cu = np.zeros(B + 1, dtype=np.int32)
        np.cumsum(lens, out=cu[1:])
        out = torch.empty((B, self.cfg["d_model"]), dtype=out_dtype or self.dtype, device=self.device)
        b0 = 0
        while b0 < B:  # passes of at most max_tokens_per_pass tokens (one pass for every realistic batch)
            b1 = int(np.searchsorted(cu, cu[b0] + self.max_tokens_per_pass, side="right")) - 1
            b1 = min(max(b1, b0 + 1), B)
            t0, t1 = int(cu[b0]), int(cu[b1])
            ids_d, cu_d = self._stage(packed[t0:t1], cu[b0 : b1 + 1] - cu[b0])
            self.encode_packed_device(ids_d, cu_d, b1 - b0, t1 - t0, int(lens[b0:b1].max()), out[b0:b1])
            b0 = b1
        return out

    def padded_workspace_bytes(self, batch: int, padded_len: int) -> int:
        return int(self._lib.rp_encode_padded_workspace_bytes(self._handle, batch, padded_len))

    def encode_padded_into(self, ids: torch.Tensor, mask: torch.Tensor, out: torch.Tensor, meta: torch.Tensor,
                           ws: Optional[torch.Tensor] = None) -> None:
        """The bare ``rp_encode_padded`` launch sequence on caller-owned buffers (int64 device ids / mask [B, L],
        out [B, d_model], meta int32 [4], optional workspace): allocation-free, so it can be captured in a hipGraph
        (reprover_amd/single_query.py)."""
        B, L = ids.shape
        assert ids.dtype == torch.int64 and mask.dtype == torch.int64 and ids.is_contiguous() and mask.is_contiguous()
        if ws is None:
            ws = self._workspace(self.padded_workspace_bytes(B, L))
        out_dt = _lib.RP_DT_BF16 if out.dtype == torch.bfloat16 else _lib.RP_DT_F32
        with torch.cuda.device(self.device):
            _lib.check(
                self._lib.rp_encode_padded(self._handle, _lib.ptr(ids), _lib.ptr(mask), B, L, out.data_ptr(), out_dt,
                                           _lib.ptr(meta), _lib.ptr(ws), ws.numel(), _lib.current_stream()),
                "rp_encode_padded",
            )

    def take_pending(self) -> list:
        """Hand the unread verdict words (device int32 [4] each) to a caller that brings them to the host itself
        (``PremiseRetriever.predict_step`` copies them along with the search result); check with ``check_verdict``."""
        metas, self._pending_meta = self._pending_meta, []
        return metas

    @staticmethod
    def check_verdict(meta_host) -> None:
        if int(meta_host[2]) != 0:
            raise ValueError("attention_mask must be right-padded (1s then 0s) with at least one token per row, "
                             "as the tokenizer produces")

    def raise_pending(self) -> None:
        """Read the verdicts of the ``encode_padded`` calls issued since the last check (synchronises)."""
        metas, self._pending_meta = self._pending_meta, []
        for m in metas:
            if int(m.cpu()[2]) != 0:
                raise ValueError("attention_mask must be right-padded (1s then 0s) with at least one token per row, "
                                 "as the tokenizer produces")
