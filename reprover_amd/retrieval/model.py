"""PremiseRetriever on MI355X: drop-in for the inference surface of the reference's
``retrieval/model.py::PremiseRetriever`` (lean-dojo/ReProver).

Same names, argument meaning and error behaviour as the reference for:
``load_hf`` (model.py:52-66), ``load_corpus`` (:68-85), ``embedding_size`` (:87-90), ``_encode``
(:92-114), ``reindex_corpus`` (:183-210), ``retrieve`` (:338-375), the predict hooks' bodies
(:274-336), the training loss ``forward`` (:116-140) and the training hooks ``on_fit_start`` / ``training_step`` /
``on_train_batch_end`` / ``configure_optimizers`` (:146-181).  There is no autograd graph: ``training_step`` runs the
forward AND the hand-written backward (reprover_amd/train.py, csrc/rp_train.hip) and leaves every parameter's gradient
in the trainer's flat buffer; the optimizer object ``configure_optimizers`` returns applies the clipped AdamW update.

What is different underneath: the encoder forward, pooling, similarity, masking and top-k all
run in hand-written HIP kernels behind the C ABI of libreprover_hip.so; premises are encoded as
packed variable-length sequences (no padding, no dense [B,H,L,L] mask), the corpus matrix stays
resident in HBM in bf16, and nothing falls back to the CPU.
"""
from __future__ import annotations

import os
import pickle
from typing import Any, Dict, List, Optional, Tuple, Union

import torch

from ..common import Context, Corpus, Fp8Index, IndexedCorpus, Pos, Premise, load_index, zip_strict
from ..encoder import HipT5Encoder
from ..tokenizer import ByT5Tokenizer


class _TrainerOptimizer:
    """What ``configure_optimizers`` hands out as "optimizer": ``step()`` = gradient clipping + one AdamW update over the
    trainer's flat buffers + refresh of the bf16 compute copies; ``zero_grad()`` is a no-op (the backward overwrites)."""

    def __init__(self, trainer, model):
        self.trainer, self.model = trainer, model

    def step(self) -> None:
        self.trainer.gradient_clip_val = self.model.gradient_clip_val
        self.trainer.optimizer_step()
        self.model.embeddings_staled = True
        self.model._drop_derived()

    def zero_grad(self, set_to_none: bool = True) -> None:
        pass

    @property
    def param_groups(self):
        return [{"lr": self.trainer.current_lr()}]


class _TrainerSchedule:
    """``get_constant_schedule_with_warmup``: the factor is a function of the optimizer steps taken, which the trainer
    counts itself, so ``step()`` has nothing to advance."""

    def __init__(self, trainer):
        self.trainer = trainer

    def step(self) -> None:
        pass

    def get_last_lr(self):
        return [self.trainer.current_lr()]


class PremiseRetriever:
    def __init__(
        self,
        model_name: Union[str, HipT5Encoder],
        lr: float = 0.0,
        warmup_steps: int = 0,
        max_seq_len: int = 2048,
        num_retrieved: int = 100,
        device: Union[str, torch.device] = "cuda",
        dtype: torch.dtype = torch.bfloat16,
        index_dtype: str = "bf16",
    ) -> None:
        self.lr = lr
        self.warmup_steps = warmup_steps
        self.num_retrieved = num_retrieved
        self.max_seq_len = max_seq_len
        self.tokenizer = ByT5Tokenizer()
        if isinstance(model_name, HipT5Encoder):
            self.encoder = model_name
        else:
            self.encoder = HipT5Encoder.from_pretrained(model_name, device, dtype)
        self.corpus: Optional[Corpus] = None
        self.corpus_embeddings: Optional[torch.Tensor] = None
        self.embeddings_staled = True
        self._predict_outputs: List[Dict[str, Any]] = []
        self.logged_metrics: Dict[str, List[float]] = {}  # what validation_step logs (name -> [weighted sum, weight])
        self.frozen = False
        self._predict_pending = None  # (batch, PendingSearch) of the last predict_step, finished lazily
        # "bf16": search the embedding matrix as it is (the reference's behaviour).  "fp8": search an
        # e4m3 copy with per-row scales (BASELINE.json configs[4]); ``corpus_embeddings`` stays what the
        # reference exposes, the quantised copy is derived from it lazily.
        assert index_dtype in ("bf16", "fp8")
        self.index_dtype = index_dtype
        self._fp8_index: Optional[Fp8Index] = None
        self._fp8_source: Optional[torch.Tensor] = None  # the tensor object the e4m3 copy was derived from
        self._attached = False  # corpus_embeddings is another process's memory (attach_index): never written here
        # Multi-GPU predict (BASELINE.json configs[2]; the reference replicates the whole index per rank):
        # when set, ``on_predict_start`` encodes only this rank's row shard and ``predict_step`` merges the
        # per-rank top-k lists through one all-gather (reprover_amd/dist.py).  Set by retrieval/main.py
        # under torch.distributed.run.
        self.shard_index_over_ranks = False
        self.index_shard = None
        self.shard_group = None  # what the sharded step gathers over: None = torch's default group, or a dist.HipComm
        # Single-state retrieve() as one hipGraph replay (reprover_amd/single_query.py); set False to launch the
        # kernels one by one as the batch paths do.
        self.use_graphs = True
        self._single_query = None
        # predict_step gathers consecutive HOST batches until this many states are queued and runs them as ONE GPU pass
        # (the reference's eval batch is 64 states; a 256-state pass costs 10 % less GPU time per state).  0 = every
        # batch is its own pass.  Records keep their order; they are complete when ``predict_step_outputs`` is read.
        self.predict_coalesce_states = 256
        # False = the reference's synchronous semantics (model.py:281-290 + common.py:323-324): predict_step finishes
        # ITS OWN batch before it returns - records appended, ValueError raised in the call that submitted the batch,
        # nothing queued or in flight afterwards.  True (default): the one-pass-deep pipeline described at predict_step.
        self.predict_pipeline = True
        self._predict_stash: List[Dict[str, Any]] = []
        # training (lazily built by training_step / configure_optimizers: fp32 masters, gradients, AdamW moments)
        self._trainer = None
        self.gradient_clip_val: Optional[float] = None  # Lightning's trainer.gradient_clip_val (confs/*.yaml: 1.0)
        # T5's dropout in training mode (HF config.dropout_rate, default 0.1 - what the reference trains with); 0 = the
        # deterministic step.  Read when the training engine is built.
        self.dropout_rate: float = float(getattr(self.encoder, "cfg", {}).get("dropout_rate", 0.1))
        self.dropout_seed: int = 3407  # the dropout masks' stream; retrieval/main.py derives it from `seed_everything`

    # -- construction (model.py:52-66) --------------------------------------------------------------
    @classmethod
    def load_hf(cls, ckpt_path: str, max_seq_len: int, device, dtype=None) -> "PremiseRetriever":
        """``dtype`` None → bf16, the reference's own choice on a capable GPU (model.py:59-64).  ``dtype`` selects
        the dtype of the embeddings handed back (``corpus_embeddings``, ``_encode``); the arithmetic is the same for
        both values - bf16 MFMA operands, fp32 accumulation and statistics, the residual stream as a bf16 plane + an int8
        extension plane (16 significant bits) - where the reference with ``dtype=float32`` multiplies fp32 operands under
        ``torch.set_float32_matmul_precision("medium")`` (model.py:26), a setting that itself licenses bf16-precision
        products inside fp32 matmuls.  ``retrieve`` / ``num_retrieved`` accept any k, as the reference does (one library call
        sorts at most 1024 keys per query; beyond that ``Corpus.get_nearest_premises`` pages through ``rp_sim_topk_after``) - on
        the unsharded index; with ``shard_index_over_ranks`` k is limited to 1024 (``dist.hip_local_topk`` raises ``ValueError``)."""
        return cls(ckpt_path, 0.0, 0, max_seq_len, 100, device=device, dtype=dtype or torch.bfloat16)

    @classmethod
    def load(cls, ckpt_path: str, device, freeze: bool = False) -> "PremiseRetriever":
        """model.py:48-50 -> common.py:414-425 ``load_checkpoint``: build the retriever from a PyTorch-Lightning checkpoint
        FILE of the reference's ``PremiseRetriever`` (what ``generation/model.py:82-84`` and the prover load).  Such a file
        is a ``torch.save``d dict: ``hyper_parameters`` (``save_hyperparameters()``: model_name, lr, warmup_steps,
        max_seq_len, num_retrieved) and ``state_dict`` whose keys carry the attribute prefix ``encoder.`` in front of the
        HuggingFace ``T5EncoderModel`` names.  The T5 geometry comes from ``<model_name>/config.json`` when
        ``model_name`` is a local directory, else from the tensor shapes (d_kv = 64; T5's default bucket / distance /
        epsilon values).  A checkpoint DIRECTORY is a DeepSpeed ZeRO checkpoint (common.py:408-411): its conversion
        script is DeepSpeed's own and out of scope here.  ``freeze`` (``model.freeze()``
        upstream: no gradients) has nothing to switch off here - there is no autograd graph - but a frozen retriever
        refuses ``training_step``."""
        if not os.path.exists(ckpt_path):
            raise FileExistsError(f"Checkpoint {ckpt_path} does not exist.")  # common.py:409-410
        if os.path.isdir(ckpt_path):
            if os.path.exists(os.path.join(ckpt_path, "zero_to_fp32.py")):
                raise NotImplementedError(
                    f"{ckpt_path} is a DeepSpeed ZeRO checkpoint: convert it with its own zero_to_fp32.py to a Lightning "
                    "checkpoint file first (the conversion is DeepSpeed code, not part of this engine)")
            return cls.load_hf(ckpt_path, 2048, device)  # a HuggingFace directory: the load_hf path
        ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
        if not isinstance(ckpt, dict) or "state_dict" not in ckpt:
            raise ValueError(f"{ckpt_path} is not a PyTorch-Lightning checkpoint (no 'state_dict')")
        hp = dict(ckpt.get("hyper_parameters") or {})
        sd = {}
        for k, v in ckpt["state_dict"].items():
            if k.startswith("encoder.") and torch.is_tensor(v):
                sd[k[len("encoder."):]] = v.detach().to(torch.float32)
        if "shared.weight" not in sd and "encoder.embed_tokens.weight" in sd:
            sd["shared.weight"] = sd["encoder.embed_tokens.weight"]
        if "shared.weight" not in sd:
            raise ValueError(f"{ckpt_path}: no T5 encoder weights under the 'encoder.' prefix")
        cfg = cls._t5_config_from(hp.get("model_name"), sd)
        enc = HipT5Encoder(cfg, sd, device, torch.bfloat16)
        model = cls(enc, lr=float(hp.get("lr", 0.0)), warmup_steps=int(hp.get("warmup_steps", 0)),
                    max_seq_len=int(hp.get("max_seq_len", 2048)), num_retrieved=int(hp.get("num_retrieved", 100)))
        model.frozen = bool(freeze)
        return model

    @staticmethod
    def _t5_config_from(model_name, sd: Dict[str, torch.Tensor]) -> Dict[str, Any]:
        import json

        n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.block."))
        q = sd["encoder.block.0.layer.0.SelfAttention.q.weight"]
        cfg = dict(vocab_size=sd["shared.weight"].shape[0], d_model=sd["shared.weight"].shape[1], d_kv=64,
                   num_heads=q.shape[0] // 64, num_layers=n_layers,
                   relative_attention_num_buckets=sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].shape[0],
                   relative_attention_max_distance=128, layer_norm_epsilon=1e-6, dropout_rate=0.1)
        if "encoder.block.0.layer.1.DenseReluDense.wi_0.weight" in sd:
            cfg.update(d_ff=sd["encoder.block.0.layer.1.DenseReluDense.wi_0.weight"].shape[0], feed_forward_proj="gated-gelu")
        else:
            cfg.update(d_ff=sd["encoder.block.0.layer.1.DenseReluDense.wi.weight"].shape[0], feed_forward_proj="relu")
        if isinstance(model_name, str) and os.path.isfile(os.path.join(model_name, "config.json")):
            with open(os.path.join(model_name, "config.json")) as fh:
                hf = json.load(fh)
            for k in ("d_kv", "num_heads", "relative_attention_num_buckets", "relative_attention_max_distance",
                      "layer_norm_epsilon", "feed_forward_proj", "dropout_rate"):
                if k in hf:
                    cfg[k] = hf[k]
        return cfg

    @classmethod
    def from_state_dict(cls, cfg: Dict, state_dict: Dict[str, torch.Tensor], max_seq_len: int, device,
                        dtype: torch.dtype = torch.bfloat16, index_dtype: str = "bf16") -> "PremiseRetriever":
        """Build from an in-memory HF-keyed state dict (synthetic weights; no checkpoint on disk)."""
        return cls(HipT5Encoder(cfg, state_dict, device, dtype), 0.0, 0, max_seq_len, 100, index_dtype=index_dtype)

    @property
    def device(self) -> torch.device:
        return self.encoder.device

    @property
    def dtype(self) -> torch.dtype:
        return self.encoder.dtype

    def eval(self) -> "PremiseRetriever":
        return self

    # -- corpus (model.py:68-85) --------------------------------------------------------------------
    def load_corpus(self, path_or_corpus: Union[str, Corpus]) -> None:
        """Associate the retriever with a corpus: a ``Corpus``, a ``corpus.jsonl`` (embeddings
        stale), a pickled ``IndexedCorpus`` with pre-computed embeddings - written by this package or by the
        reference's ``retrieval/index.py`` (its classes are mapped while unpickling: no lean_dojo / networkx
        needed) -, or a native index directory written by ``common.save_index`` / ``index.py --output-path <dir>/``."""
        self._drop_derived()
        self._attached = False
        if isinstance(path_or_corpus, Corpus):
            self.corpus = path_or_corpus
            self.corpus_embeddings = None
            self.embeddings_staled = True
            return
        path = path_or_corpus
        if os.path.isdir(path):  # native index directory (common.save_index)
            self.corpus, self.corpus_embeddings, fp8 = load_index(path, with_fp8=True)
            self.embeddings_staled = False
            if fp8 is not None and self.index_dtype == "fp8":  # persisted e4m3 form: no re-quantisation
                self._fp8_index = Fp8Index(fp8[0].to(self.device), fp8[1].to(self.device))
                self._fp8_source = self.corpus_embeddings
            return
        if path.endswith(".jsonl"):
            self.corpus = Corpus(path)
            self.corpus_embeddings = None
            self.embeddings_staled = True
        else:  # a pickled IndexedCorpus: this package's, or the REFERENCE's own (retrieval/index.py:37-40)
            from ..common import load_indexed_corpus_pickle

            self.corpus, self.corpus_embeddings = load_indexed_corpus_pickle(path)
            self.embeddings_staled = False

    def _search_operand(self):
        """What ``get_nearest_premises`` scans: the embeddings, or their e4m3 copy (rebuilt whenever
        ``corpus_embeddings`` is replaced or written)."""
        if self.index_dtype != "fp8":
            return self.corpus_embeddings
        # Keyed on the tensor OBJECT (a strong reference, so its storage cannot be recycled under the tag) and
        # dropped explicitly wherever the matrix is rewritten in place through raw pointers (_drop_derived):
        # data_ptr()/_version would both survive an allocator-recycled block filled by rp_encode_varlen.
        ver = getattr(self.corpus_embeddings, "_version", None)
        if (self._fp8_index is None or self._fp8_source is not self.corpus_embeddings
                or getattr(self, "_fp8_version", ver) != ver):  # replaced, or written in place through torch
            self._fp8_index = Fp8Index.quantize(self.corpus_embeddings, self.device)
            self._fp8_source = self.corpus_embeddings
        self._fp8_version = ver
        return self._fp8_index

    # ---- one index per GPU, shared by the worker processes on it (reprover_amd/shared_index.py) -------------
    def share_index(self):
        """Owner side: a picklable handle to this retriever's device-resident index (bf16 matrix, mask arrays and,
        when ``index_dtype`` is e4m3, the codes and scales).  Keep this retriever alive while workers use it."""
        from ..shared_index import export_index

        assert self.corpus is not None and self.corpus_embeddings is not None and not self.embeddings_staled
        if self.corpus_embeddings.device != self.device or self.corpus_embeddings.dtype != torch.bfloat16:
            self._drop_derived()
            self.corpus_embeddings = self.corpus_embeddings.to(device=self.device, dtype=torch.bfloat16).contiguous()
        fp8 = self._search_operand() if self.index_dtype != "bf16" else None
        return export_index(self.corpus_embeddings, self.corpus, fp8)

    def attach_index(self, handle, corpus: Union[str, Corpus]) -> None:
        """Worker side: use the owner's index in place (no copy).  ``corpus`` is this process's own host-side corpus
        (a ``Corpus`` or the path of the same ``corpus.jsonl`` / index directory): premise objects live on the host."""
        from ..shared_index import attach_index

        if isinstance(corpus, Corpus):
            self.corpus = corpus
        else:
            self.load_corpus(corpus)
        self._drop_derived()
        t = attach_index(handle)
        assert handle.n_premises == len(self.corpus), "the handle belongs to a different corpus"
        assert t["embeddings"].device == self.device, "attach on the GPU the owner exported from"
        self.corpus_embeddings = t["embeddings"]
        self.corpus._dev[str(self.device)] = (t["file_of"], t["end_key"])
        if handle.has_fp8:
            self._fp8_index = Fp8Index(t["fp8_codes"], t["fp8_scale"])
            self._fp8_source = self.corpus_embeddings
        self.embeddings_staled = False
        self._attached = True

    def _drop_derived(self) -> None:
        """Forget every copy derived from ``corpus_embeddings`` (e4m3 index, cached bf16 cast)."""
        from ..common import drop_cast_cache

        self._fp8_index = None
        self._fp8_source = None
        if getattr(self, "_single_query", None) is not None:
            self._single_query.clear()
        drop_cast_cache()

    @property
    def embedding_size(self) -> int:
        return self.encoder.config.hidden_size

    # -- encode (model.py:92-114) -------------------------------------------------------------------
    def _encode(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        """Unit-norm feature vectors [B, D] for right-padded token batches."""
        return self.encoder.encode_padded(input_ids, attention_mask)

    # -- training forward (model.py:116-140) --------------------------------------------------------
    def forward(self, context_ids, context_mask, pos_premise_ids, pos_premise_mask, neg_premises_ids, neg_premises_mask,
                label) -> torch.Tensor:
        """The contrastive loss of the reference's ``forward``: encode the contexts, the positive premises and
        every list of negatives (fp32 embeddings), ``similarity = context_emb @ all_premise_embs.T``,
        ``loss = F.mse_loss(similarity, label)`` - all on the GPU through ``rp_encode_padded`` and
        ``rp_contrastive_mse``; returns a 0-dim fp32 device tensor, ``self.last_similarity`` keeps the [B, P]
        matrix.  Forward only: there are no backward kernels, so the tensor carries no graph."""
        from .. import _lib
        from ..common import _workspace

        assert len(neg_premises_ids) == len(neg_premises_mask)
        f32 = torch.float32
        ctx = self.encoder.encode_padded(context_ids, context_mask, defer_check=True, out_dtype=f32)
        prem = [self.encoder.encode_padded(pos_premise_ids, pos_premise_mask, defer_check=True, out_dtype=f32)]
        for ids, mask in zip_strict(neg_premises_ids, neg_premises_mask):
            prem.append(self.encoder.encode_padded(ids, mask, defer_check=True, out_dtype=f32))
        all_prem = torch.cat(prem, dim=0).contiguous()
        B, D = ctx.shape
        P = all_prem.shape[0]
        lab = label.to(device=self.device, dtype=f32).contiguous()
        assert lab.shape == (B, P), f"label {tuple(lab.shape)} != ({B}, {P})"
        loss = torch.empty((), dtype=f32, device=self.device)
        sim = torch.empty((B, P), dtype=f32, device=self.device)
        lib = _lib.load()
        nbytes = lib.rp_contrastive_mse_workspace_bytes(B, P)
        ws = _workspace(self.device, nbytes)
        with torch.cuda.device(self.device):
            _lib.check(lib.rp_contrastive_mse(_lib.ptr(ctx), _lib.ptr(all_prem), _lib.ptr(lab), B, P, D, loss.data_ptr(),
                                              _lib.ptr(sim), _lib.ptr(ws), nbytes, _lib.current_stream()),
                       "rp_contrastive_mse")
        self.encoder.raise_pending()
        self.last_similarity = sim
        return loss

    __call__ = forward

    # -- training (model.py:146-181) ------------------------------------------------------------------
    def train_engine(self):
        """The ``HipT5Trainer`` behind ``training_step`` (built on first use from the encoder's fp32 weights); from then
        on ``self.encoder`` is the inference engine over the trainer's CURRENT weights."""
        if self._trainer is None:
            from ..train import HipT5Trainer

            sd = self.encoder.master_weights() if hasattr(self.encoder, "master_weights") else None
            if sd is None:
                raise RuntimeError("training needs the encoder's fp32 weights (build the retriever from a checkpoint "
                                   "or a state dict)")
            self._trainer = HipT5Trainer(self.encoder.cfg, sd, self.device, lr=self.lr, warmup_steps=self.warmup_steps,
                                         gradient_clip_val=self.gradient_clip_val, out_dtype=self.encoder.dtype,
                                         dropout_rate=self.dropout_rate, dropout_seed=self.dropout_seed)
            self.encoder = self._trainer.encoder
            self._drop_derived()
        return self._trainer

    def on_fit_start(self, corpus: Optional[Corpus] = None) -> None:
        """model.py:146-153: take the datamodule's corpus; its embeddings are stale from here on."""
        if corpus is not None:
            self.corpus = corpus
        self._drop_derived()
        self.corpus_embeddings = None
        self.embeddings_staled = True

    # -- validation (model.py:212-268) ----------------------------------------------------------------
    def log(self, name: str, value, on_epoch: bool = True, sync_dist: bool = True, batch_size: int = 1, **_) -> None:
        """Lightning's ``self.log(..., on_epoch=True, batch_size=n)``: the epoch value is the mean over the steps weighted
        by ``batch_size``.  Kept in ``logged_metrics`` (name -> [weighted sum, weight]); read with ``epoch_metrics()``."""
        acc = self.logged_metrics.setdefault(name, [0.0, 0.0])
        acc[0] += float(value) * batch_size
        acc[1] += batch_size

    def epoch_metrics(self) -> Dict[str, float]:
        return {k: (v[0] / v[1] if v[1] else 0.0) for k, v in self.logged_metrics.items()}

    def on_validation_start(self, eval_batch_size: Optional[int] = None) -> None:
        """model.py:212-213: re-index the corpus with the datamodule's eval batch size (argument, or
        ``self.trainer.datamodule.eval_batch_size`` when a Lightning-style trainer object is attached)."""
        if eval_batch_size is None:
            tr = getattr(self, "trainer", None)
            eval_batch_size = tr.datamodule.eval_batch_size if tr is not None else 64
        self.logged_metrics = {}
        self.reindex_corpus(eval_batch_size)

    def validation_step(self, batch: Dict[str, Any], batch_idx: int = 0) -> None:
        """model.py:215-268: retrieve for the batch, then Recall@1..k (in %) and MRR over the examples that have positive
        premises, logged with ``batch_size = num_with_premises`` as upstream."""
        from .evaluate import recall_and_mrr

        context_emb = self._encode(batch["context_ids"], batch["context_mask"])
        assert not self.embeddings_staled
        retrieved, _ = self.corpus.get_nearest_premises(self.corpus_embeddings, batch["context"], context_emb,
                                                        self.num_retrieved)
        if not any(len(p) for p in batch["all_pos_premises"]):
            return  # (upstream logs the mean of an empty list - NaN with batch_size 0 - here: nothing)
        recall, mrr, n = recall_and_mrr(batch["all_pos_premises"], retrieved, self.num_retrieved)
        for j in range(self.num_retrieved):
            self.log(f"Recall@{j + 1}_val", recall[j], on_epoch=True, sync_dist=True, batch_size=n)
        self.log("MRR", mrr, on_epoch=True, sync_dist=True, batch_size=n)

    def training_step(self, batch: Dict[str, Any], _=None) -> torch.Tensor:
        """model.py:155-167: the contrastive loss of ``forward`` on a training batch (``collate`` with is_train) - and,
        since nothing records a graph here, its backward: on return every parameter's gradient lies in
        ``train_engine().grads`` (all five encodes run as ONE packed pass).  Returns the loss (0-dim fp32 device tensor)."""
        if getattr(self, "frozen", False):
            raise RuntimeError("this retriever was loaded with freeze=True")
        tr = self.train_engine()
        groups = [(batch["context_ids"], batch["context_mask"]), (batch["pos_premise_ids"], batch["pos_premise_mask"])]
        groups += list(zip_strict(batch["neg_premises_ids"], batch["neg_premises_mask"]))
        loss, sim = tr.contrastive_step(groups, batch["label"])
        self.last_similarity = sim
        return loss

    def on_train_batch_end(self, outputs=None, batch=None, _=None) -> None:
        """model.py:169-171."""
        self.embeddings_staled = True

    def configure_optimizers(self) -> Dict[str, Any]:
        """common.py:381-405 (``get_optimizers``): AdamW(lr) under the constant schedule with linear warm-up, the
        scheduler stepping once per optimizer step - in the reference's dictionary shape.  ``optimizer.step()`` applies
        Lightning's ``gradient_clip_val`` (when set) and the update to the gradients the last ``training_step`` left."""
        tr = self.train_engine()
        return {"optimizer": _TrainerOptimizer(tr, self), "lr_scheduler": {"scheduler": _TrainerSchedule(tr), "interval": "step"}}

    def encode_texts(self, texts: List[str], out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Tokenise + encode without materialising padding (the packed form the engine consumes)."""
        ids, cu = self.tokenizer.packed(texts, self.max_seq_len)
        return self.encoder.encode_packed(ids, cu, out)

    # -- index (model.py:183-210) -------------------------------------------------------------------
    @torch.no_grad()
    def reindex_corpus(self, batch_size: int) -> None:
        """Re-encode every premise of the corpus if the embeddings are stale.

        The reference walks the corpus ``batch_size`` premises at a time, padding each batch to its
        longest member.  Encoding a premise does not depend on its batch, so
        here ``batch_size`` only bounds the host-side serialise/tokenise step; the GPU receives
        packed chunks sized by tokens.  Row i of ``corpus_embeddings`` is premise i, as in the
        reference."""
        if not self.embeddings_staled:
            return
        if self._attached:
            raise RuntimeError("this retriever searches an index owned by another process (attach_index); "
                               "re-index in the owner")
        N = len(self.corpus.all_premises)
        self._drop_derived()
        self.corpus_embeddings = torch.zeros(N, self.embedding_size, dtype=self.encoder.dtype, device=self.device)
        step = max(int(batch_size), 4096)
        for i in range(0, N, step):
            chunk = self.corpus.all_premises[i : i + step]
            self.encode_texts([p.serialize() for p in chunk], out=self.corpus_embeddings[i : i + len(chunk)])
        self.embeddings_staled = False

    # -- prediction (model.py:274-336) --------------------------------------------------------------
    def on_predict_start(self, corpus: Optional[Corpus] = None, eval_batch_size: Optional[int] = None) -> None:
        """model.py:274-279.  The reference's hook takes no arguments and reads ``self.trainer.datamodule.corpus`` /
        ``.eval_batch_size``: called that way (a Lightning-style ``trainer`` attribute attached to the object) it does the
        same here; this package's own driver (``retrieval/main.py``) passes the two values."""
        if corpus is None or eval_batch_size is None:
            dm = getattr(getattr(self, "trainer", None), "datamodule", None)
            if dm is None:
                raise TypeError("on_predict_start() needs (corpus, eval_batch_size) or an attached trainer.datamodule")
            corpus = dm.corpus if corpus is None else corpus
            eval_batch_size = dm.eval_batch_size if eval_batch_size is None else eval_batch_size
        self.corpus = corpus
        self._drop_derived()
        self.corpus_embeddings = None
        self.embeddings_staled = True
        self.predict_step_outputs = []
        if self.shard_index_over_ranks:
            import torch.distributed as dist

            from .. import dist as rdist

            bounds = rdist.shard_bounds(rdist.premise_token_counts(corpus, self.max_seq_len), dist.get_world_size())
            self.index_shard = rdist.IndexShard(corpus, bounds, dist.get_rank(), self.device)
            rdist.reindex_shard(self, self.index_shard)
            if os.environ.get("RP_COMM") == "abi" and self.shard_group is None and self.device.type == "cuda":
                # the step's collective through the library's own RCCL communicator (rp_comm_*) instead of torch's
                self.shard_group = rdist.HipComm.from_torch_group(device=self.device)
            if self.index_dtype == "fp8":
                self.index_shard.quantize()  # the sharded search then scans the e4m3 form, like the single-GPU one
            return
        self.reindex_corpus(eval_batch_size)

    # ``predict_step_outputs`` (model.py:274, 314-327) - the reference's attribute.  Here the records of the batches still
    # queued or in flight are completed lazily: reading the attribute (or on_predict_epoch_end) launches and finishes them.
    @property
    def predict_step_outputs(self) -> List[Dict[str, Any]]:
        self._launch_predict(self._take_predict_stash())
        self._finish_pending_predict()
        return self._predict_outputs

    @predict_step_outputs.setter
    def predict_step_outputs(self, value: List[Dict[str, Any]]) -> None:
        self._predict_pending = None
        self._predict_stash = []
        self._predict_outputs = value

    def _take_predict_stash(self) -> List[Dict[str, Any]]:
        stash, self._predict_stash = self._predict_stash, []
        return stash

    def predict_step(self, batch: Dict[str, Any], _=None) -> None:
        """model.py:281-327 as a software pipeline.  (1) Consecutive host batches are gathered until
        ``predict_coalesce_states`` states are queued and then encoded and searched as ONE GPU pass (a row's result does
        not depend on its pass; the pass size is what the GPU's efficiency depends on).  (2) A pass is only ENQUEUED here
        (encode, masked top-k, copy to pinned memory); the records of the PREVIOUS pass are completed afterwards, so the
        host-side mapping - and whatever the caller does between calls: collating the next batch - overlaps GPU work.
        Consequence: a ``ValueError`` (fewer than k accessible premises; a mask that is not right-padded) surfaces later
        than in the reference - at the latest when ``predict_step_outputs`` is read or the epoch ends.

        ``predict_pipeline = False`` is the STRICT form: no gathering, no overlap - the call encodes, searches and maps
        its own batch and raises the reference's ``ValueError`` itself, exactly where model.py:281-290 does."""
        if not self.predict_pipeline:
            self._launch_predict(self._take_predict_stash())  # (batches queued before the switch was flipped)
            self._finish_pending_predict()
            self._launch_predict([batch])
            self._finish_pending_predict()
            return
        host = not batch["context_ids"].is_cuda and not batch["context_mask"].is_cuda
        if host and self.predict_coalesce_states > len(batch["context"]):
            self._predict_stash.append(batch)
            if sum(len(b["context"]) for b in self._predict_stash) >= self.predict_coalesce_states:
                self._launch_predict(self._take_predict_stash())
            return
        self._launch_predict(self._take_predict_stash())  # (keeps the order of the records)
        self._launch_predict([batch])

    def _launch_predict(self, batches: List[Dict[str, Any]]) -> None:
        if not batches:
            return
        if len(batches) == 1:  # launch-only encode (device form: mask -> lengths -> packed ids on the device)
            context_emb = self.encoder.encode_padded(batches[0]["context_ids"], batches[0]["context_mask"], defer_check=True)
        else:
            context_emb = self.encoder.encode_padded_many([(b["context_ids"], b["context_mask"]) for b in batches])
        contexts = [c for b in batches for c in b["context"]]
        if self.index_shard is not None:  # every rank holds the batch; the index is row-sharded: same pipeline
            from ..dist import launch_sharded_nearest_premises

            launched = launch_sharded_nearest_premises(self.index_shard, contexts, context_emb, self.num_retrieved,
                                                       group=self.shard_group, also_copy=self.encoder.take_pending())
        else:
            assert not self.embeddings_staled
            launched = self.corpus.launch_nearest_premises(
                self._search_operand(), contexts, context_emb, self.num_retrieved,
                also_copy=self.encoder.take_pending(),  # the encode's right-padding verdict travels with the result
            )
        previous, self._predict_pending = self._predict_pending, (batches, launched)
        self._finish_pending_predict(previous)

    def _finish_pending_predict(self, pending="current") -> None:
        if pending == "current":
            pending, self._predict_pending = self._predict_pending, None
        if pending is None:
            return
        batches, launched = pending
        retrieved_premises, scores = launched.finish()
        for verdict in launched.extra_host:
            self.encoder.check_verdict(verdict)
        lo = 0
        for batch in batches:
            hi = lo + len(batch["context"])
            self._append_predictions(batch, retrieved_premises[lo:hi], scores[lo:hi])
            lo = hi

    def _append_predictions(self, batch: Dict[str, Any], retrieved_premises, scores) -> None:
        for url, commit, file_path, full_name, start, tactic_idx, ctx, pos_premises, premises, s in zip_strict(
            batch["url"], batch["commit"], batch["file_path"], batch["full_name"], batch["start"],
            batch["tactic_idx"], batch["context"], batch["all_pos_premises"], retrieved_premises, scores,
        ):
            self._predict_outputs.append(
                {
                    "url": url,
                    "commit": commit,
                    "file_path": file_path,
                    "full_name": full_name,
                    "start": start,
                    "tactic_idx": tactic_idx,
                    "context": ctx,
                    "all_pos_premises": pos_premises,
                    "retrieved_premises": premises,
                    "scores": s,
                }
            )

    def on_predict_epoch_end(self, log_dir: Optional[str]) -> None:
        if log_dir is not None:
            path = os.path.join(log_dir, "predictions.pickle")
            with open(path, "wb") as oup:
                pickle.dump(self.predict_step_outputs, oup)  # (reading the attribute completes the last batch)
        self.predict_step_outputs.clear()

    # -- single query (model.py:338-375) ------------------------------------------------------------
    @torch.no_grad()
    def retrieve(
        self, state: str, file_name: str, theorem_full_name: str, theorem_pos: Pos, k: int
    ) -> Tuple[List[Premise], List[float]]:
        """Retrieve ``k`` premises from the corpus using ``state`` as the query."""
        self.reindex_corpus(batch_size=32)
        ctx = Context(file_name, theorem_full_name, theorem_pos, state)
        if (self.use_graphs and self.index_dtype == "bf16" and self.corpus_embeddings.device == self.device
                and self.corpus_embeddings.dtype == torch.bfloat16 and k <= 1024):
            # the prover's hot call: one H2D copy, one graph replay (encode + masked top-k), one D2H copy
            if self._single_query is None:
                from ..single_query import SingleQueryCache

                self._single_query = SingleQueryCache()
            got = self._single_query.retrieve(self, ctx, k)
            if got is not None:
                return got  # (None: the captured two-pass search overflowed its candidate list - redo it below)
        context_emb = self.encode_texts([ctx.serialize()])
        if self.corpus_embeddings.device != context_emb.device or self.corpus_embeddings.dtype != torch.bfloat16:
            # a pickled index arrives as fp32 on the CPU (index.py:37-40): move + cast once
            keep = self._fp8_index if self._fp8_source is self.corpus_embeddings else None  # e4m3 form from disk
            self._drop_derived()
            self.corpus_embeddings = self.corpus_embeddings.to(device=context_emb.device, dtype=torch.bfloat16)
            if keep is not None:
                self._fp8_index, self._fp8_source = keep, self.corpus_embeddings
        retrieved_premises, scores = self.corpus.get_nearest_premises(self._search_operand(), [ctx], context_emb, k)
        assert len(retrieved_premises) == len(scores) == 1
        return retrieved_premises[0], scores[0]
