"""Multi-GPU retrieval: the premise-embedding matrix row-sharded over the GPUs of one node.

The reference has no multi-GPU retrieval at all (every Lightning rank / Ray actor re-indexes and
holds the full corpus: retrieval/model.py:274-279, prover/proof_search.py:438-447).  Here, one
process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI):

  * re-index: premises are independent, so rank r encodes only rows [lo_r, hi_r) — no exchange at
    all.  Bounds are chosen on the cumulative serialized byte length so every rank encodes about the
    same number of tokens (encode cost ~ tokens), while shards stay contiguous row ranges and global
    premise ids stay ``lo_r + local row``.
  * retrieve: every rank scans its shard for ALL queries of the step with the accessibility mask
    applied before selection (``rp_sim_topk`` with ``id_offset = lo_r``) and writes its lists straight into ONE
    packed block ``[scores f32 [B,k] | ids i32 [B,k] | counts i32 [B]]``; ONE all-gather of that block
    (B*(2k+1)*4 bytes per rank: latency-bound) and a k-way merge that reads the receive buffer as it lies
    (``rp_topk_merge_strided``: no copies between the collective and the merge).  Masked top-k is a decomposable
    reduction, so the result is exactly the single-GPU result.

The collective plumbing is backend-agnostic; the two compute steps are injectable so that the
world_size-2 ``gloo`` tests on CPU can drive the same code with the oracle as the checker.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .common import Context, Corpus, _workspace, as_bf16_matrix

TopK = Tuple[torch.Tensor, torch.Tensor, torch.Tensor]  # ids int32 [B,k], scores f32 [B,k], counts int32 [B]


def shard_bounds(weights: Sequence[int], world: int) -> np.ndarray:
    """Contiguous row ranges with (nearly) equal total weight: int64 [world+1], bounds[0] = 0,
    bounds[world] = N.  ``weights[i]`` = token count (serialized byte length + 1) of premise i."""
    w = np.asarray(weights, dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(w)])
    targets = cum[-1] * np.arange(1, world) / world
    inner = np.searchsorted(cum, targets, side="left")
    b = np.concatenate([[0], inner, [len(w)]]).astype(np.int64)
    return np.maximum.accumulate(b)


def premise_token_counts(corpus: Corpus, max_seq_len: int) -> np.ndarray:
    return np.fromiter(
        (min(len(p.serialize().encode("utf-8")) + 1, max_seq_len) for p in corpus.all_premises),
        dtype=np.int64,
        count=len(corpus),
    )


class IndexShard:
    """Rows [lo, hi) of the corpus embedding matrix on this rank, with the matching slices of the
    per-premise accessibility arrays."""

    def __init__(self, corpus: Corpus, bounds: np.ndarray, rank: int, device: torch.device):
        self.corpus = corpus
        self.bounds = np.asarray(bounds, dtype=np.int64)
        self.rank = rank
        self.lo, self.hi = int(bounds[rank]), int(bounds[rank + 1])
        self.device = device
        self.embeddings: Optional[torch.Tensor] = None  # [hi-lo, D] on `device`
        self.file_of = torch.from_numpy(corpus.file_of[self.lo : self.hi].copy()).to(device)
        self.end_key = torch.from_numpy(corpus.end_key[self.lo : self.hi].copy()).to(device)

    def __len__(self) -> int:
        return self.hi - self.lo

    def quantize(self) -> None:
        """Derive the e4m3 form of this shard's rows (``PremiseRetriever(index_dtype="fp8")`` under sharding)."""
        from .common import Fp8Index

        self.fp8 = Fp8Index.quantize(self.embeddings, self.device)


def reindex_shard(retriever, shard: IndexShard) -> None:
    """This rank's part of ``reindex_corpus`` (retrieval/model.py:183-210): encode premises
    [lo, hi) into ``shard.embeddings``.  No communication."""
    prem = retriever.corpus.all_premises[shard.lo : shard.hi]
    out = torch.zeros(len(prem), retriever.embedding_size, dtype=retriever.encoder.dtype, device=retriever.device)
    step = 4096
    for i in range(0, len(prem), step):
        chunk = prem[i : i + step]
        retriever.encode_texts([p.serialize() for p in chunk], out=out[i : i + len(chunk)])
    shard.embeddings = out


def hip_local_topk(shard: IndexShard, batch_context: Sequence[Context], query_emb: torch.Tensor, k: int,
                   flags: Optional[int] = None) -> TopK:
    """Masked top-k of all queries against this rank's rows (global ids), on the GPU.  ``shard.fp8`` (an
    ``Fp8Index`` of the shard's rows, set by ``IndexShard.quantize``) switches the scan to the e4m3 form.
    ``flags`` None: the two-pass plan, redone densely if a candidate list overflowed (reads the counts: synchronises);
    an explicit value: exactly that plan, launch-only (the caller inspects the counts later)."""
    lib = _lib.load()
    dev = query_emb.device
    from .common import MAX_K_PER_CALL

    if k > MAX_K_PER_CALL:  # (the single-GPU search pages through rp_sim_topk_after; the sharded step has one exchange)
        raise ValueError(f"sharded retrieval serves k <= {MAX_K_PER_CALL} per query (got k = {k}): one rp_sim_topk call per "
                         f"shard, one packed exchange; use the unsharded index for a deeper ranking")
    fp8 = getattr(shard, "fp8", None)
    if fp8 is not None:
        return _hip_local_topk_fp8(shard, fp8, batch_context, query_emb, k, flags)
    E = as_bf16_matrix(shard.embeddings, dev)
    Q = as_bf16_matrix(query_emb, dev)
    B, D = Q.shape
    d_bits, d_own, d_qk = shard.corpus.device_query_masks(batch_context, dev)  # 12 B per query uploaded, pinned + async
    out_i, out_s, out_c = packed_topk_buffers(B, k, dev)

    def scan(flags: int) -> None:
        nbytes = lib.rp_sim_topk_workspace_bytes(B, len(shard), D, k, flags)
        ws = _workspace(dev, nbytes)
        _lib.check(
            lib.rp_sim_topk(
                _lib.ptr(Q), _lib.ptr(E), B, len(shard), D, _lib.ptr(shard.file_of), _lib.ptr(shard.end_key),
                _lib.ptr(d_bits), shard.corpus.num_files, _lib.ptr(d_own), _lib.ptr(d_qk), shard.lo, k, flags,
                _lib.ptr(out_s), _lib.ptr(out_i), _lib.ptr(out_c), _lib.ptr(ws), nbytes, _lib.current_stream(),
            ),
            "rp_sim_topk",
        )

    if flags is not None:
        scan(flags)
        return out_i, out_s, out_c
    scan(_lib.RP_TOPK_AUTO)
    # The C ABI reserves out_count = -1 for "candidate list overflow: call again with RP_TOPK_DENSE"
    # (include/reprover_hip.h).  The list is bounded (plan_sim: 8 k stride keys, at least 8192), so adversarial scores or a
    # sample without a bound CAN overflow it; the contract is honoured here as in Corpus.get_nearest_premises: a rank whose
    # list overflowed must not drop out of the merge (merge_keys treats a negative count as "no candidates").
    if bool((out_c < 0).any()):
        scan(_lib.RP_TOPK_DENSE)
    return out_i, out_s, out_c


def _hip_local_topk_fp8(shard: IndexShard, fp8, batch_context: Sequence[Context], query_emb: torch.Tensor, k: int,
                        only_flags: Optional[int] = None) -> TopK:
    from .common import Fp8Index

    lib = _lib.load()
    dev = query_emb.device
    Q = Fp8Index.quantize(query_emb)
    B, D = Q.shape
    d_bits, d_own, d_qk = shard.corpus.device_query_masks(batch_context, dev)
    out_i, out_s, out_c = packed_topk_buffers(B, k, dev)
    for flags in ((_lib.RP_TOPK_AUTO, _lib.RP_TOPK_DENSE) if only_flags is None else (only_flags,)):
        nbytes = lib.rp_sim_topk_workspace_bytes(B, len(shard), D, k, flags)
        ws = _workspace(dev, nbytes)
        _lib.check(
            lib.rp_sim_topk_fp8(
                _lib.ptr(Q.codes), _lib.ptr(Q.scale), _lib.ptr(fp8.codes), _lib.ptr(fp8.scale), B, len(shard), D,
                _lib.ptr(shard.file_of), _lib.ptr(shard.end_key), _lib.ptr(d_bits), shard.corpus.num_files, _lib.ptr(d_own),
                _lib.ptr(d_qk), shard.lo, k, flags, _lib.ptr(out_s), _lib.ptr(out_i), _lib.ptr(out_c), _lib.ptr(ws),
                nbytes, _lib.current_stream(),
            ),
            "rp_sim_topk_fp8",
        )
        if only_flags is not None or not bool((out_c < 0).any()):  # -1 = candidate overflow (reserved by the ABI): redo densely
            break
    return out_i, out_s, out_c


def packed_topk_buffers(B: int, k: int, device) -> TopK:
    """(ids [B,k], scores [B,k], counts [B]) as views of ONE int32 block laid out [scores | ids | counts]: what a rank
    contributes to the all-gather, written in place by ``rp_sim_topk``."""
    block = torch.empty(B * (2 * k + 1), dtype=torch.int32, device=device)
    return (block[B * k : 2 * B * k].view(B, k), block[: B * k].view(torch.float32).view(B, k), block[2 * B * k :])


def _packed_block(ids: torch.Tensor, scores: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
    """The [scores | ids | counts] block the three tensors are views of (``packed_topk_buffers``); built by one
    concatenation when they are not (injected compute steps of the CPU tests)."""
    B, k = ids.shape
    n = B * (2 * k + 1)
    st = scores.untyped_storage()
    if (st.data_ptr() == ids.untyped_storage().data_ptr() == counts.untyped_storage().data_ptr() and st.nbytes() == 4 * n
            and scores.storage_offset() == 0 and ids.storage_offset() == B * k and counts.storage_offset() == 2 * B * k
            and scores.is_contiguous() and ids.is_contiguous()):
        return torch.empty(0, dtype=torch.int32, device=ids.device).set_(st, 0, (n,))
    return torch.cat([scores.contiguous().view(torch.int32).reshape(-1), ids.to(torch.int32).reshape(-1),
                      counts.to(torch.int32).reshape(-1)])


def hip_merge(ids: torch.Tensor, scores: torch.Tensor, counts: torch.Tensor) -> TopK:
    """Merge [R, B, k] per-rank lists into the global top-k on the GPU.  The inputs may be strided views of an
    all-gather's receive buffer (rank stride > B*k): they are read in place (``rp_topk_merge_strided``)."""
    lib = _lib.load()
    R, B, k = scores.shape
    dev = scores.device
    out_s = torch.empty((B, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((B, k), dtype=torch.int32, device=dev)
    out_c = torch.empty((B,), dtype=torch.int32, device=dev)
    nbytes = lib.rp_topk_merge_workspace_bytes(R, B, k)
    ws = _workspace(dev, nbytes)
    rs = scores.stride(0)
    if (R > 1 and rs != B * k and rs == ids.stride(0) == counts.stride(0) and scores.stride()[1:] == (k, 1)
            and ids.stride()[1:] == (k, 1) and counts.stride(1) == 1):
        _lib.check(
            lib.rp_topk_merge_strided(scores.data_ptr(), ids.data_ptr(), counts.data_ptr(), rs, R, B, k, _lib.ptr(out_s),
                                      _lib.ptr(out_i), _lib.ptr(out_c), _lib.ptr(ws), nbytes, _lib.current_stream()),
            "rp_topk_merge_strided",
        )
        return out_i, out_s, out_c
    _lib.check(
        lib.rp_topk_merge(_lib.ptr(scores.contiguous()), _lib.ptr(ids.contiguous()), _lib.ptr(counts.contiguous()),
                          R, B, k, _lib.ptr(out_s), _lib.ptr(out_i), _lib.ptr(out_c), _lib.ptr(ws), nbytes,
                          _lib.current_stream()),
        "rp_topk_merge",
    )
    return out_i, out_s, out_c


class HipComm:
    """An RCCL communicator held by libreprover_hip itself (``rp_comm_*``, include/reprover_hip.h): the step's collective
    without ``torch.distributed`` in the data path.  Accepted wherever this module takes ``group``.  The 128-byte id made by
    rank 0 (``HipComm.unique_id()``) reaches the other ranks by any channel the caller has; ``from_torch_group`` uses an
    existing torch group (gloo is enough) for that one broadcast."""

    def __init__(self, unique_id: bytes, rank: int, world: int, device: Optional[torch.device] = None):
        import ctypes as C

        assert len(unique_id) == 128
        self.rank, self.world = int(rank), int(world)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:  # "cuda" -> the current device, so that the device checks
            self.device = torch.device("cuda", torch.cuda.current_device())  # below compare like with like (cuda:0 != cuda)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().rp_comm_init(C.c_char_p(unique_id), self.rank, self.world, C.byref(handle)), "rp_comm_init")
        self._handle = handle

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C

        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().rp_comm_unique_id(buf), "rp_comm_unique_id")
        return buf.raw

    @classmethod
    def from_torch_group(cls, group=None, device: Optional[torch.device] = None) -> "HipComm":
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(box[0], rank, world, device)

    def close(self) -> None:
        if self._handle is not None:
            handle, self._handle = self._handle, None
            _lib.check(_lib.load().rp_comm_destroy(handle), "rp_comm_destroy")

    def __del__(self):  # best effort; close() is the explicit form
        try:
            self.close()
        except Exception:
            pass

    def all_gather_stack(self, t: torch.Tensor) -> torch.Tensor:
        t = t.contiguous()
        assert t.is_cuda and (t.numel() * t.element_size()) % 4 == 0
        assert t.device == self.device, f"tensor on {t.device}, communicator on {self.device}"
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        with torch.cuda.device(self.device):  # the stream handed over must be the communicator device's current stream
            _lib.check(_lib.load().rp_comm_allgather(self._handle, t.data_ptr(), out.data_ptr(), t.numel() * t.element_size(),
                                                     _lib.current_stream()), "rp_comm_allgather")
        return out

    def allgather_topk(self, block: torch.Tensor, Bt: int, k: int, q0: int, B: int):
        """``rp_allgather_topk``: (ids [B,k], scores [B,k], counts [B]) of queries [q0, q0 + B) merged over the ranks, and
        the receive buffer [world, Bt (2k + 1)] (its counts section carries every rank's overflow verdict)."""
        lib = _lib.load()
        dev = block.device
        assert block.dtype == torch.int32 and block.is_contiguous() and block.numel() == Bt * (2 * k + 1)
        recv = torch.empty((self.world, block.numel()), dtype=torch.int32, device=dev)
        out_s = torch.empty((B, k), dtype=torch.float32, device=dev)
        out_i = torch.empty((B, k), dtype=torch.int32, device=dev)
        out_c = torch.empty((B,), dtype=torch.int32, device=dev)
        nbytes = lib.rp_topk_merge_workspace_bytes(self.world, B, k)
        ws = _workspace(dev, nbytes)
        assert dev == self.device, f"block on {dev}, communicator on {self.device}"
        with torch.cuda.device(self.device):
            _lib.check(lib.rp_allgather_topk(self._handle, block.data_ptr(), recv.data_ptr(), Bt, k, q0, B, _lib.ptr(out_s),
                                             _lib.ptr(out_i), _lib.ptr(out_c), _lib.ptr(ws), nbytes, _lib.current_stream()),
                       "rp_allgather_topk")
        return out_i, out_s, out_c, recv


def _world_rank(group) -> Tuple[int, int]:
    return (group.world, group.rank) if isinstance(group, HipComm) else (dist.get_world_size(group), dist.get_rank(group))


def all_gather_stack(t: torch.Tensor, group=None) -> torch.Tensor:
    """[world, *t.shape]: one all-gather (RCCL ncclAllGather on GPUs - through torch.distributed or, when ``group`` is a
    ``HipComm``, through the library's own communicator; gloo on CPU)."""
    if isinstance(group, HipComm):
        return group.all_gather_stack(t)
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    else:
        dist.all_gather(list(out.unbind(0)), t.contiguous(), group=group)
    return out


def gather_shards(local_rows: torch.Tensor, bounds: np.ndarray, group=None) -> torch.Tensor:
    """Assemble the full [N, D] matrix from every rank's rows [bounds[r], bounds[r+1]) — the one
    exchange of a multi-GPU re-index that has to persist a single index file (index.py).  Shards
    differ in length (they balance tokens, not rows), so each rank contributes a block padded to the
    longest shard and the padding is dropped after ONE all-gather."""
    world, rank = _world_rank(group)
    bounds = np.asarray(bounds, dtype=np.int64)
    assert len(bounds) == world + 1 and local_rows.shape[0] == bounds[rank + 1] - bounds[rank]
    longest = int(np.diff(bounds).max())
    block = torch.zeros((longest,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype, device=local_rows.device)
    block[: local_rows.shape[0]] = local_rows
    stacked = all_gather_stack(block, group)
    return torch.cat([stacked[r, : int(bounds[r + 1] - bounds[r])] for r in range(world)], dim=0)


def sharded_nearest_premise_ids(
    shard: IndexShard,
    batch_context: Sequence[Context],
    query_emb: torch.Tensor,
    k: int,
    group=None,
    local_topk: Callable[..., TopK] = hip_local_topk,
    merge: Callable[..., TopK] = hip_merge,
) -> TopK:
    """Global masked top-k for a batch of queries that every rank holds (e.g. after an all-gather of
    the per-rank query embeddings).  Returns identical tensors on every rank."""
    ids, scores, counts = local_topk(shard, batch_context, query_emb, k)
    B = ids.shape[0]
    if isinstance(group, HipComm) and merge is hip_merge:  # collective + merge as ONE call of the C ABI
        return group.allgather_topk(_packed_block(ids, scores, counts), B, k, 0, B)[:3]
    g = all_gather_stack(_packed_block(ids, scores, counts), group)  # THE collective of the step: [world, B (2k + 1)]
    g_scores = g[:, : B * k].view(torch.float32).view(-1, B, k)      # views of the receive buffer, rank stride B (2k + 1)
    g_ids = g[:, B * k : 2 * B * k].view(-1, B, k)
    g_counts = g[:, 2 * B * k :]
    return merge(g_ids, g_scores, g_counts)


_default_sliced_staging: dict = {}  # send / receive buffers of sliced_exchange_merge calls that pass no `staging`


def sliced_exchange_merge(ids: torch.Tensor, scores: torch.Tensor, counts: torch.Tensor, group=None,
                          merge: Callable[..., TopK] = hip_merge, staging: Optional[dict] = None) -> TopK:
    """The result exchange of a step in which every rank OWNS a slice of the queries (weak scaling: query q belongs to rank
    q // Bq, Bq = B_all / world) but has scanned its index shard for all of them: rank r needs, from every rank, only the
    lists of its own Bq queries.  ONE ``all_to_all`` of per-destination slices ``[scores Bq k | ids Bq k | counts Bq]``
    (Bq (2k + 1) 4 bytes per peer) instead of the packed all-gather, which delivers every rank's lists for ALL queries -
    world x the bytes, of which a rank merges 1 / world.  The merge then reads the receive buffer as
    it lies (rank stride Bq (2k + 1)).  Returns (ids [Bq, k], scores, counts) of this rank's queries.
    ``staging``: an optional dict the caller keeps between steps (send / receive buffers are allocated once)."""
    if isinstance(group, HipComm):  # (before anything is packed or allocated)
        raise NotImplementedError("the library's communicator carries the all-gather form only (rp_allgather_topk)")
    world, rank = _world_rank(group)
    B_all, k = ids.shape
    assert B_all % world == 0, "every rank owns the same number of queries"
    Bq = B_all // world
    blk = Bq * (2 * k + 1)
    dev = ids.device
    st = staging if staging is not None else _default_sliced_staging  # (a caller without its own dict still allocates once)
    key = ("sliced", world, Bq, k, str(dev))
    if key not in st:
        st[key] = (torch.empty((world, blk), dtype=torch.int32, device=dev), torch.empty((world, blk), dtype=torch.int32, device=dev))
    send, recv = st[key]
    # per-destination slices: three strided copies (the scan writes [scores | ids | counts] over ALL queries)
    send[:, : Bq * k].copy_(scores.contiguous().view(torch.int32).view(world, Bq * k))
    send[:, Bq * k : 2 * Bq * k].copy_(ids.to(torch.int32).contiguous().view(world, Bq * k))
    send[:, 2 * Bq * k :].copy_(counts.to(torch.int32).contiguous().view(world, Bq))
    if dist.get_backend(group) == "nccl" or not send.is_cuda:
        dist.all_to_all_single(recv, send, group=group)
    else:  # functional runs of the N > 1 path on one GPU over gloo: gloo's all-to-all takes host tensors
        r_h = torch.empty(send.shape, dtype=torch.int32)
        dist.all_to_all_single(r_h, send.cpu(), group=group)
        recv.copy_(r_h)
    g_scores = recv[:, : Bq * k].view(torch.float32).view(world, Bq, k)
    g_ids = recv[:, Bq * k : 2 * Bq * k].view(world, Bq, k)
    g_counts = recv[:, 2 * Bq * k :]
    return merge(g_ids, g_scores, g_counts)


def sharded_get_nearest_premises(shard: IndexShard, batch_context: List[Context], query_emb: torch.Tensor, k: int,
                                 group=None, **kw):
    """Sharded drop-in for ``Corpus.get_nearest_premises`` (common.py:299-326): same return value and
    the same ``ValueError`` when a query has fewer than ``k`` accessible premises in the whole corpus."""
    ids, scores, counts = sharded_nearest_premise_ids(shard, batch_context, query_emb, k, group, **kw)
    if bool((counts.cpu() < k).any()):
        raise ValueError
    prem = shard.corpus.all_premises
    return [[prem[i] for i in row] for row in ids.cpu().tolist()], scores.cpu().tolist()


_rank_count_pool: dict = {}  # (world, B) -> pinned int32 [world, B] buffers (at most 4 each)


class PendingShardedSearch:
    """A sharded search whose merged result is on its way to pinned host memory (``launch_sharded_nearest_premises``):
    the sharded counterpart of ``common.PendingSearch``, so that ``predict_step`` pipelines both the same way."""

    def __init__(self, shard, batch_context, query_emb, k, group, host, done, extra_host=()):
        self.shard, self.batch_context, self.query_emb, self.k, self.group = shard, list(batch_context), query_emb, k, group
        self.host, self.done, self.extra_host = host, done, list(extra_host)

    def finish(self):
        """Wait for the copy, map ids to ``Premise`` objects; ``ValueError`` as the reference (common.py:323-324)."""
        self.done.synchronize()
        ids_h, scores_h, counts_h, rank_counts_h = self.host
        k = self.k
        if bool((rank_counts_h < 0).any()):
            # some rank's candidate list overflowed (out_count = -1, reserved by the ABI).  Every rank holds the same
            # gathered counts, so every rank takes this branch: redo the step with the dense plan, synchronously.
            ids, scores, counts = sharded_nearest_premise_ids(
                self.shard, self.batch_context, self.query_emb, k, self.group,
                local_topk=lambda s, c, q, kk: hip_local_topk(s, c, q, kk, _lib.RP_TOPK_DENSE))
            ids_h, scores_h, counts_h = ids.cpu(), scores.cpu(), counts.cpu()
        self.query_emb = None
        try:
            if bool((counts_h < k).any()):
                raise ValueError
            ids_l, scores_l = ids_h.tolist(), scores_h.tolist()
        finally:  # the pinned buffers go back to their pools (bounded: common._PINNED_POOL_SHAPES)
            from .common import _pinned_pool

            p3 = _pinned_pool.get((len(self.batch_context), k))
            if p3 is not None and len(p3) < 4:
                p3.append(self.host[:3])
            pc = _rank_count_pool.get(tuple(self.host[3].shape))
            if pc is not None and len(pc) < 4:
                pc.append(self.host[3])
        prem = self.shard.corpus.all_premises
        return [[prem[i] for i in row] for row in ids_l], scores_l


def launch_sharded_nearest_premises(shard: IndexShard, batch_context: List[Context], query_emb: torch.Tensor, k: int,
                                    group=None, also_copy: Sequence[torch.Tensor] = ()) -> PendingShardedSearch:
    """First half of the sharded ``get_nearest_premises``: local masked top-k (two-pass plan, launch-only), the packed
    all-gather, the merge and the asynchronous copy of the merged lists (plus every rank's counts, for the overflow
    contract) to pinned host memory are enqueued; nothing waits.  ``finish()`` is the second half."""
    ids, scores, counts = hip_local_topk(shard, batch_context, query_emb, k, _lib.RP_TOPK_AUTO)
    B = ids.shape[0]
    if isinstance(group, HipComm):  # collective + merge as one call of the C ABI (rp_allgather_topk)
        m_ids, m_scores, m_counts, g = group.allgather_topk(_packed_block(ids, scores, counts), B, k, 0, B)
        world = g.shape[0]
    else:
        g = all_gather_stack(_packed_block(ids, scores, counts), group)
        world = g.shape[0]
        m_ids, m_scores, m_counts = hip_merge(g[:, B * k : 2 * B * k].view(world, B, k),
                                              g[:, : B * k].view(torch.float32).view(world, B, k), g[:, 2 * B * k :])
    from .common import _pinned_result_buffers

    pool = _rank_count_pool.setdefault((world, B), [])
    host = _pinned_result_buffers(B, k) + ((pool.pop() if pool else torch.empty((world, B), dtype=torch.int32).pin_memory()),)
    host[0].copy_(m_ids, non_blocking=True)
    host[1].copy_(m_scores, non_blocking=True)
    host[2].copy_(m_counts, non_blocking=True)
    host[3].copy_(g[:, 2 * B * k :], non_blocking=True)
    from .common import _pinned_small

    extra = [_pinned_small(t).copy_(t, non_blocking=True) for t in also_copy]
    done = torch.cuda.Event()
    done.record(torch.cuda.current_stream(query_emb.device))
    return PendingShardedSearch(shard, batch_context, query_emb, k, group, host, done, extra)
