"""Single-state retrieval as ONE hipGraph replay.

The prover calls ``PremiseRetriever.retrieve`` once per search-tree node (reference
prover/tactic_generator.py:286-292): B = 1, ~100-1000 tokens, ~90 kernel launches (7 per encoder layer + the
scan's five).  Launched one by one the path is host-bound: most of those kernels run for 3-10 us, a launch costs
the host 3-4 us, and every call allocates, copies three small mask arrays and synchronises twice.  Here the whole
device side - ``rp_encode_padded`` (mask -> lengths -> packed ids -> encoder -> pooled unit vector) followed by
``rp_sim_topk`` (masked similarity + exact top-k over the resident index) - is captured once per
(token-length bucket, k) into a hipGraph on static buffers:

    one pinned host buffer   [ids | mask | q_key | own_file]  --one H2D copy-->  device twin
    graph replay             (the token count stays on the device: the padded entry point needs no host value; the
                              accessibility bits are built from the resident import closure: rp_build_file_bits)
    one D2H copy             [scores | ids | count | meta]                --one synchronisation

PyTorch supplies the stream-capture plumbing (``torch.cuda.CUDAGraph``) and the memory; every captured node is one
of the engine's own launches.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .common import Context, Corpus, Premise
from .tokenizer import encode_one

BUCKETS = (128, 256, 512, 1024, 2048)


class SingleQueryGraph:
    """Static buffers + captured graph for one (padded length, k)."""

    def __init__(self, encoder, corpus: Corpus, E: torch.Tensor, L: int, k: int):
        assert E.dtype == torch.bfloat16 and E.is_cuda and E.is_contiguous()
        self.encoder, self.corpus, self.E, self.L, self.k = encoder, corpus, E, L, k
        dev = E.device
        lib = _lib.load()
        F, N, D = corpus.num_files, E.shape[0], E.shape[1]
        # ---- input block: int64 ids [L], int64 mask [L], int64 q_key, int32 own_file (+ pad)
        self._n_in = 2 * L * 8 + 16
        self.h_in = torch.zeros(self._n_in, dtype=torch.uint8).pin_memory()
        self.d_in = torch.zeros(self._n_in, dtype=torch.uint8, device=dev)
        hv, dv = self.h_in.numpy(), self.d_in
        self.h_ids = hv[: L * 8].view(np.int64)
        self.h_mask = hv[L * 8 : 2 * L * 8].view(np.int64)
        self.h_qk = hv[2 * L * 8 : 2 * L * 8 + 8].view(np.int64)
        self.h_own = hv[2 * L * 8 + 8 : 2 * L * 8 + 12].view(np.int32)
        self.d_ids = dv[: L * 8].view(torch.int64).view(1, L)
        self.d_mask = dv[L * 8 : 2 * L * 8].view(torch.int64).view(1, L)
        self.d_qk = dv[2 * L * 8 : 2 * L * 8 + 8].view(torch.int64)
        self.d_own = dv[2 * L * 8 + 8 : 2 * L * 8 + 12].view(torch.int32)
        self.d_bits = torch.zeros((F, 1), dtype=torch.int32, device=dev)  # built by the graph from the resident closure
        self.reach = corpus.device_reach(dev)
        # ---- output block: int32 meta [4] (16-byte aligned), f32 scores [k], int32 ids [k], int32 count
        self._n_out = 16 + 8 * k + 4
        self.d_out = torch.zeros(self._n_out, dtype=torch.uint8, device=dev)
        self.h_out = torch.zeros(self._n_out, dtype=torch.uint8).pin_memory()
        self.d_meta = self.d_out[:16].view(torch.int32)
        self.d_scores = self.d_out[16 : 16 + 4 * k].view(torch.float32).view(1, k)
        self.d_topk = self.d_out[16 + 4 * k : 16 + 8 * k].view(torch.int32).view(1, k)
        self.d_count = self.d_out[16 + 8 * k :].view(torch.int32)
        ho = self.h_out.numpy()
        self.h_meta = ho[:16].view(np.int32)
        self.h_scores, self.h_topk = ho[16 : 16 + 4 * k].view(np.float32), ho[16 + 4 * k : 16 + 8 * k].view(np.int32)
        self.h_count = ho[16 + 8 * k :].view(np.int32)
        # ---- scratch owned by the graph (the captured nodes hold raw pointers)
        self.q = torch.zeros((1, D), dtype=torch.bfloat16, device=dev)
        self.ws_enc = torch.empty(encoder.padded_workspace_bytes(1, L), dtype=torch.uint8, device=dev)
        self._ws_sim_bytes = int(lib.rp_sim_topk_workspace_bytes(1, N, D, k, 0))
        self.ws_sim = torch.empty(self._ws_sim_bytes, dtype=torch.uint8, device=dev)
        self.file_of, self.end_key = corpus._device_arrays(dev)
        self.h_mask[0] = 1
        self.h_ids[0] = 1  # a valid one-token sequence for the warm-up / capture runs
        self._launch()  # warm-up outside capture (first-use attribute calls, allocator quiescence)
        torch.cuda.current_stream(dev).synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._launch(copy_in=False)

    def _launch(self, copy_in: bool = True) -> None:
        lib = _lib.load()
        if copy_in:
            self.d_in.copy_(self.h_in, non_blocking=True)
        self.encoder.encode_padded_into(self.d_ids, self.d_mask, self.q, self.d_meta, self.ws_enc)
        N, D = self.E.shape
        with torch.cuda.device(self.E.device):
            _lib.check(lib.rp_build_file_bits(_lib.ptr(self.reach), self.corpus.num_files, _lib.ptr(self.d_own), 1,
                                              _lib.ptr(self.d_bits), _lib.current_stream()), "rp_build_file_bits")
            _lib.check(
                lib.rp_sim_topk(_lib.ptr(self.q), _lib.ptr(self.E), 1, N, D, _lib.ptr(self.file_of), _lib.ptr(self.end_key),
                                _lib.ptr(self.d_bits), self.corpus.num_files, _lib.ptr(self.d_own), _lib.ptr(self.d_qk), 0,
                                self.k, 0, _lib.ptr(self.d_scores), _lib.ptr(self.d_topk), _lib.ptr(self.d_count),
                                _lib.ptr(self.ws_sim), self._ws_sim_bytes, _lib.current_stream()),
                "rp_sim_topk",
            )

    def run(self, ids: np.ndarray, ctx: Context) -> Tuple[np.ndarray, np.ndarray, int]:
        """ids: int32 token ids incl. EOS (len <= L).  Returns (premise ids [k], scores [k], count)."""
        n = ids.size
        self.h_ids[:n] = ids
        self.h_ids[n:] = 0
        self.h_mask[:n] = 1
        self.h_mask[n:] = 0
        self.h_own[0] = self.corpus._index[ctx.path]  # KeyError for a file outside the corpus, as query_masks
        self.h_qk[0] = ctx.theorem_pos.key()
        self.d_in.copy_(self.h_in, non_blocking=True)
        self.graph.replay()
        self.h_out.copy_(self.d_out, non_blocking=True)
        torch.cuda.current_stream(self.E.device).synchronize()
        if self.h_meta[2] != 0:  # the device-side check of rp_encode_padded (cannot fire for a tokenised state)
            raise ValueError(f"rp_encode_padded rejected the state: meta={self.h_meta.tolist()} for {n} tokens")
        return self.h_topk.copy(), self.h_scores.copy(), int(self.h_count[0])


class SingleQueryCache:
    """Graphs of one retriever, keyed by (bucket, k); dropped whenever the index or the corpus object changes."""

    def __init__(self) -> None:
        self._graphs: Dict[Tuple[int, int], SingleQueryGraph] = {}
        self._key = None

    def clear(self) -> None:
        self._graphs.clear()
        self._key = None

    def retrieve(self, retriever, ctx: Context, k: int) -> Optional[Tuple[List[Premise], List[float]]]:
        E, corpus = retriever.corpus_embeddings, retriever.corpus
        key = (id(E), id(corpus))
        if key != self._key:
            self.clear()
            self._key = key
            self._pin = (E, corpus)  # the ids stay unique while these are alive
        ids = encode_one(ctx.serialize(), retriever.max_seq_len)
        # token-length bucket: the graph is captured for L padded tokens, the real count stays on the device
        L = next((b for b in BUCKETS if b >= ids.size), ((ids.size + 127) // 128) * 128)
        g = self._graphs.get((L, k))
        if g is None:
            g = self._graphs[(L, k)] = SingleQueryGraph(retriever.encoder, corpus, E, L, k)
        top, scores, count = g.run(ids, ctx)
        if count < 0:  # candidate-list overflow (out_count = -1): the caller repeats launch by launch, dense plan
            return None
        if count < k:
            raise ValueError  # fewer than k accessible premises (common.py:323-324)
        prem = corpus.all_premises
        return [prem[i] for i in top.tolist()], scores.tolist()
