"""Host-side corpus data model of the retrieval path, MI355X-native.

Mirrors the interface of the reference's ``common.py`` (lean-dojo/ReProver) for the classes the
retrieval hot path touches — ``Pos``, ``Context``, ``Premise``, ``File``, ``Corpus``,
``IndexedCorpus`` — with the same names, argument meaning and error behaviour, but a different
representation underneath:

  * the import DAG's transitive closure is a packed bit matrix built in one topological sweep
    (the reference uses ``networkx.transitive_closure_dag``, common.py:215);
  * accessibility (common.py:280-289) is kept in array form — ``file_of[i]``, ``end_key[i]`` per
    premise and one bit per (file, query) — which is what the HIP scan kernel consumes, so the
    per-query Python ``PremiseSet`` walk of common.py:312-324 never happens;
  * ``get_nearest_premises`` (common.py:299-326) runs on the GPU through ``rp_sim_topk``.
"""
from __future__ import annotations

import json
import pickle
import re
from dataclasses import dataclass, field
from typing import Any, Dict, Generator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

MARK_START_SYMBOL = "<a>"  # common.py:25
MARK_END_SYMBOL = "</a>"  # common.py:26

POS_COL_BITS = 20  # key = (line_nb << 20) | column_nb, monotone in the lexicographic Pos order


def remove_marks(s: str) -> str:
    """common.py:29-31."""
    return s.replace(MARK_START_SYMBOL, "").replace(MARK_END_SYMBOL, "")


class Pos(tuple):
    """Stand-in for ``lean_dojo.Pos`` (absent here): ``Pos(line_nb, column_nb)``, iterable,
    hashable, totally ordered lexicographically."""

    __slots__ = ()

    def __new__(cls, line_nb: int, column_nb: int):
        return super().__new__(cls, (int(line_nb), int(column_nb)))

    def __getnewargs__(self):  # picklable (an IndexedCorpus pickle holds every premise's positions)
        return (self[0], self[1])

    @property
    def line_nb(self) -> int:
        return self[0]

    @property
    def column_nb(self) -> int:
        return self[1]

    def key(self) -> int:
        assert 0 <= self[1] < (1 << POS_COL_BITS) and self[0] >= 0
        return (self[0] << POS_COL_BITS) | self[1]

    def __repr__(self) -> str:
        return f"Pos(line_nb={self[0]}, column_nb={self[1]})"


@dataclass(unsafe_hash=True)
class Context:
    """A retrieval query (common.py:34-56)."""

    path: str
    theorem_full_name: str
    theorem_pos: Pos = field(compare=False)
    state: str

    def __post_init__(self) -> None:
        assert isinstance(self.path, str)
        assert isinstance(self.theorem_full_name, str)
        assert isinstance(self.theorem_pos, Pos)
        assert (
            isinstance(self.state, str)
            and "⊢" in self.state
            and MARK_START_SYMBOL not in self.state
            and MARK_END_SYMBOL not in self.state
        )

    def serialize(self) -> str:
        return self.state


_WS_LOOKBEHIND = r"(?<=\s)«?"


@dataclass(unsafe_hash=True)
class Premise:
    """A retrievable document (common.py:59-106)."""

    path: str
    full_name: str
    start: Pos = field(repr=False)
    end: Pos = field(repr=False, compare=False)
    code: str = field(compare=False)

    def __post_init__(self) -> None:
        assert isinstance(self.path, str)
        assert isinstance(self.full_name, str)
        assert isinstance(self.start, Pos) and isinstance(self.end, Pos) and self.start <= self.end
        assert isinstance(self.code, str) and self.code != ""

    def serialize(self) -> str:
        """Text fed to the encoder: the code with the premise's own name wrapped in <a>…</a>
        (byte-identical to common.py:93-106, fixture tests/golden/g2_serialize.json)."""
        tagged = MARK_START_SYMBOL + self.full_name + MARK_END_SYMBOL
        code = self.code.replace("_root_." + self.full_name, tagged)
        name = self.full_name
        while True:
            # the reference interpolates the (unescaped) dotted suffix into the pattern
            out = re.sub(_WS_LOOKBEHIND + name + "»?", tagged, code)
            if out != code:
                return out
            dot = name.find(".")
            if dot < 0:
                return code
            name = name[dot + 1 :]


class PremiseSet:
    """Premises indexed by (path, full_name) (common.py:109-138)."""

    def __init__(self) -> None:
        self.path2premises: Dict[str, Dict[str, Premise]] = {}

    def __iter__(self) -> Generator[Premise, None, None]:
        for group in self.path2premises.values():
            yield from group.values()

    def add(self, p: Premise) -> None:
        self.path2premises.setdefault(p.path, {})[p.full_name] = p

    def update(self, premises: Sequence[Premise]) -> None:
        for p in premises:
            self.add(p)

    def __contains__(self, p: Premise) -> bool:
        return p.full_name in self.path2premises.get(p.path, ())

    def __len__(self) -> int:
        return sum(len(g) for g in self.path2premises.values())


@dataclass(frozen=True)
class File:
    """A Lean file and the premises it defines (common.py:141-178)."""

    path: str
    premises: List[Premise] = field(repr=False, compare=False)

    @classmethod
    def from_data(cls, file_data: Dict[str, Any]) -> "File":
        path = file_data["path"]
        kept = []
        for rec in file_data["premises"]:
            name = rec["full_name"]
            if name is None or "user__.n" in name or rec["code"] == "":
                continue  # ill-formed (common.py:160-164)
            if name.startswith("[") and name.endswith("]"):
                continue  # mutual definitions (common.py:165-167)
            kept.append(Premise(path, name, Pos(*rec["start"]), Pos(*rec["end"]), rec["code"]))
        return cls(path, kept)

    @property
    def is_empty(self) -> bool:
        return self.premises == []


MAX_K_PER_CALL = 1024  # keys one rp_sim_topk call selects and sorts per query (include/reprover_hip.h)


class Corpus:
    """The retrieval corpus: a DAG of files whose premises can be retrieved (common.py:181-326)."""

    all_premises: List[Premise]

    def __init__(self, jsonl_path: Optional[str], _arrays: Optional[Dict[str, np.ndarray]] = None,
                 _files: Optional[Sequence[Tuple["File", Sequence[str]]]] = None) -> None:
        """``_arrays``: the array form persisted in a native index directory (``save_index``); when given,
        the import closure and the per-premise arrays are taken from it instead of being rebuilt.
        ``_files``: (File, imported paths) pairs in place of a ``corpus.jsonl`` (``from_files``)."""
        self._files: List[File] = []
        self._index: Dict[str, int] = {}
        direct: List[List[int]] = []
        self.all_premises = []

        def records():
            if _files is not None:
                yield from _files
                return
            with open(jsonl_path) as fh:
                for line in fh:
                    data = json.loads(line)
                    yield File.from_data(data), data["imports"]

        for f, imports in records():
            assert f.path not in self._index  # common.py:204
            deps = []
            for imp in imports:
                assert imp in self._index  # imports must precede importers (common.py:211)
                deps.append(self._index[imp])
            self._index[f.path] = len(self._files)
            self._files.append(f)
            direct.append(deps)
            self.all_premises.extend(f.premises)
        if _arrays is not None:
            self._adopt_arrays(_arrays)
        else:
            self._build_arrays(direct)

    @classmethod
    def from_files(cls, files: Sequence[Tuple["File", Sequence[str]]]) -> "Corpus":
        """A corpus from (File, paths it imports) pairs in topological order; the imports may be direct or already
        transitive (the closure is idempotent)."""
        return cls(None, _files=list(files))

    def _adopt_arrays(self, a: Dict[str, np.ndarray]) -> None:
        F, N = len(self._files), len(self.all_premises)
        self._reach = np.ascontiguousarray(a["reach"]).view(np.uint64).reshape(F, (F + 63) // 64)
        self.file_of = np.ascontiguousarray(a["file_of"], dtype=np.int32)
        self.end_key = np.ascontiguousarray(a["end_key"], dtype=np.int64)
        self._file_start = np.ascontiguousarray(a["file_start"], dtype=np.int64)
        if self.file_of.shape != (N,) or self.end_key.shape != (N,) or self._file_start.shape != (F + 1,):
            raise ValueError("index arrays do not match corpus.jsonl (premise / file counts differ)")
        self._dev = {}

    def index_arrays(self) -> Dict[str, np.ndarray]:
        """The array form to persist: transitive import closure (bit g of row f: f imports g), file index and
        name-group end key of every premise, first premise of every file."""
        return {"reach": self._reach.view(np.int64), "file_of": self.file_of, "end_key": self.end_key,
                "file_start": self._file_start}

    # -- array form -----------------------------------------------------------------------------
    def _build_arrays(self, direct: List[List[int]]) -> None:
        F = len(self._files)
        W = (F + 63) // 64
        reach = np.zeros((F, W), dtype=np.uint64)  # bit g of row f: f imports g (transitively)
        for f, deps in enumerate(direct):  # files arrive in topological order
            row = reach[f]
            for g in deps:
                row |= reach[g]
                row[g >> 6] |= np.uint64(1) << np.uint64(g & 63)
        self._reach = reach
        N = len(self.all_premises)
        self.file_of = np.zeros(N, dtype=np.int32)
        self.end_key = np.zeros(N, dtype=np.int64)
        self._file_start = np.zeros(F + 1, dtype=np.int64)
        i = 0
        for f, fl in enumerate(self._files):
            self._file_start[f] = i
            # PremiseSet membership is by (path, full_name): a later duplicate of a name is
            # "accessible" as soon as ANY same-named premise of the file is (common.py:129-132),
            # so each premise carries the smallest end key of its name group.
            best: Dict[str, int] = {}
            for p in fl.premises:
                k = p.end.key()
                if k < best.get(p.full_name, 1 << 62):
                    best[p.full_name] = k
            for p in fl.premises:
                self.file_of[i] = f
                self.end_key[i] = best[p.full_name]
                i += 1
        self._file_start[F] = i
        self._dev: Dict[str, torch.Tensor] = {}

    def __getstate__(self):  # pickled inside IndexedCorpus: never carry device tensors along
        state = dict(self.__dict__)
        state["_dev"] = {}
        return state

    # -- reference interface --------------------------------------------------------------------
    def _get_file(self, path: str) -> File:
        return self._files[self._index[path]]

    def __len__(self) -> int:
        return len(self.all_premises)

    def __contains__(self, path: str) -> bool:
        return path in self._index

    def __getitem__(self, idx: int) -> Premise:
        return self.all_premises[idx]

    @property
    def files(self) -> List[File]:
        return list(self._files)

    @property
    def num_files(self) -> int:
        return len(self._files)

    def _reach_ids(self, f: int) -> np.ndarray:
        bits = np.unpackbits(self._reach[f].view(np.uint8), bitorder="little")[: len(self._files)]
        return np.flatnonzero(bits)

    def get_dependencies(self, path: str) -> List[str]:
        """Direct and indirect imports of ``path`` (common.py:241-243)."""
        return [self._files[g].path for g in self._reach_ids(self._index[path])]

    def get_premises(self, path: str) -> List[Premise]:
        return self._get_file(path).premises

    def num_premises(self, path: str) -> int:
        return len(self.get_premises(path))

    def locate_premise(self, path: str, pos: Pos) -> Optional[Premise]:
        """common.py:253-262."""
        for p in self.get_premises(path):
            if p.start <= pos <= p.end:
                return p
        return None

    def _get_imported_premises(self, path: str) -> List[Premise]:
        out: List[Premise] = []
        for g in self._reach_ids(self._index[path]):
            out.extend(self._files[g].premises)
        return out

    def get_accessible_premises(self, path: str, pos: Pos) -> PremiseSet:
        """Premises of (transitively) imported files plus those ending at or before ``pos`` in
        the same file (common.py:280-289)."""
        s = PremiseSet()
        for p in self.get_premises(path):
            if p.end <= pos:
                s.add(p)
        s.update(self._get_imported_premises(path))
        return s

    def accessible_mask(self, path: str, pos: Pos) -> np.ndarray:
        """bool [N]: the array form of ``p in get_accessible_premises(path, pos)``."""
        f = self._index[path]
        bits = np.unpackbits(self._reach[f].view(np.uint8), bitorder="little")[: len(self._files)].astype(bool)
        return bits[self.file_of] | ((self.file_of == f) & (self.end_key <= pos.key()))

    def get_accessible_premise_indexes(self, path: str, pos: Pos) -> List[int]:
        """common.py:291-297 (index form: uses each premise's own end, not its name group's)."""
        f = self._index[path]
        reach = set(self._reach_ids(f).tolist())
        return [
            i
            for i, p in enumerate(self.all_premises)
            if (self.file_of[i] == f and p.end <= pos) or int(self.file_of[i]) in reach
        ]

    # -- GPU search ------------------------------------------------------------------------------
    def query_masks(self, batch_context: Sequence[Context]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Per-batch accessibility operands of ``rp_sim_topk``:
        (file_bits_t uint32 [F, ceil(B/32)], own_file int32 [B], q_key int64 [B])."""
        B, F = len(batch_context), len(self._files)
        own = np.fromiter((self._index[c.path] for c in batch_context), dtype=np.int32, count=B)  # KeyError
        qk = np.fromiter((c.theorem_pos.key() for c in batch_context), dtype=np.int64, count=B)
        rows = np.unpackbits(self._reach[own].view(np.uint8), axis=1, bitorder="little")[:, :F]  # [B, F]
        words = (B + 31) // 32
        padded = np.zeros((F, words * 32), dtype=np.uint8)
        padded[:, :B] = rows.T
        bits_t = np.packbits(padded, axis=1, bitorder="little").view(np.uint32).reshape(F, words)
        return np.ascontiguousarray(bits_t), own, qk

    def query_keys(self, batch_context: Sequence[Context]) -> Tuple[np.ndarray, np.ndarray]:
        """(own_file int32 [B], q_key int64 [B]): everything a search uploads per query (12 bytes)."""
        B = len(batch_context)
        own = np.fromiter((self._index[c.path] for c in batch_context), dtype=np.int32, count=B)  # KeyError
        qk = np.fromiter((c.theorem_pos.key() for c in batch_context), dtype=np.int64, count=B)
        return own, qk

    def device_reach(self, device: torch.device) -> torch.Tensor:
        """The transitive import closure (bit g of row f: f imports g) resident on ``device``: int64 view of
        uint64 [F, ceil(F / 64)] - 3.1 MB at 5,000 files, uploaded once."""
        key = "reach:" + str(device)
        if key not in self._dev:
            self._dev[key] = torch.from_numpy(np.ascontiguousarray(self._reach).view(np.int64)).to(device)
        return self._dev[key]

    def device_query_masks(self, batch_context: Sequence[Context], device: torch.device,
                           out_bits: Optional[torch.Tensor] = None):
        """The accessibility operands of ``rp_sim_topk`` ON the device: (file_bits_t int32 [F, ceil(B/32)], own_file
        int32 [B], q_key int64 [B]).  The host sends own_file and q_key (one pinned asynchronous copy of 12 bytes per
        query); the bit matrix is built there from the resident closure (``rp_build_file_bits``).  Same bits as
        ``query_masks``."""
        own, qk = self.query_keys(batch_context)
        d_own, d_qk = _upload_query_keys(own, qk, device)
        B, F = len(batch_context), len(self._files)
        bits = out_bits if out_bits is not None else torch.empty((F, (B + 31) // 32), dtype=torch.int32, device=device)
        with torch.cuda.device(device):
            _lib.check(_lib.load().rp_build_file_bits(_lib.ptr(self.device_reach(device)), F, _lib.ptr(d_own), B,
                                                      _lib.ptr(bits), _lib.current_stream()), "rp_build_file_bits")
        return bits, d_own, d_qk

    def _device_arrays(self, device: torch.device) -> Tuple[torch.Tensor, torch.Tensor]:
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (
                torch.from_numpy(self.file_of).to(device),
                torch.from_numpy(self.end_key).to(device),
            )
        return self._dev[key]

    def nearest_premise_ids(
        self,
        premise_embeddings: "torch.Tensor | Fp8Index",
        batch_context: Sequence[Context],
        batch_context_emb: torch.Tensor,
        k: int,
        dense: bool = False,
    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Device-side search: (ids int32 [B,k], scores f32 [B,k], counts int32 [B]) on the GPU.  One library call sorts
        at most ``MAX_K_PER_CALL`` = 1024 keys per query; a larger k (the reference accepts any, common.py:299-326) is
        served page by page - ``rp_sim_topk_after`` continues the ranking behind the last entry of the page before -
        and the pages concatenate to exactly the single-call answer (the order is total: ids break ties)."""
        if k > MAX_K_PER_CALL:
            return self._nearest_premise_ids_paged(premise_embeddings, batch_context, batch_context_emb, k, dense)
        lib = _lib.load()
        dev = batch_context_emb.device
        if dev.type != "cuda":
            raise _lib.HipLibraryError("nearest-premise search runs on the GPU only (no CPU fallback)")
        fp8 = isinstance(premise_embeddings, Fp8Index)
        if fp8:
            E = premise_embeddings
            assert E.device == dev, "the e4m3 index must live on the query device"
            Q = Fp8Index.quantize(batch_context_emb)
        else:
            E = as_bf16_matrix(premise_embeddings, dev)
            Q = as_bf16_matrix(batch_context_emb, dev)
        B, D = Q.shape
        N = E.shape[0]
        assert N == len(self.all_premises) and E.shape[1] == D
        file_of, end_key = self._device_arrays(dev)
        d_bits, d_own, d_qk = self.device_query_masks(batch_context, dev)
        out_s = torch.empty((B, k), dtype=torch.float32, device=dev)
        out_i = torch.empty((B, k), dtype=torch.int32, device=dev)
        out_c = torch.empty((B,), dtype=torch.int32, device=dev)
        flags = _lib.RP_TOPK_DENSE if dense else _lib.RP_TOPK_AUTO
        ws_bytes = lib.rp_sim_topk_workspace_bytes(B, N, D, k, flags)
        ws = _workspace(dev, ws_bytes)
        if fp8:
            _lib.check(
                lib.rp_sim_topk_fp8(
                    _lib.ptr(Q.codes), _lib.ptr(Q.scale), _lib.ptr(E.codes), _lib.ptr(E.scale), B, N, D,
                    _lib.ptr(file_of), _lib.ptr(end_key), _lib.ptr(d_bits), len(self._files), _lib.ptr(d_own),
                    _lib.ptr(d_qk), 0, k, flags, _lib.ptr(out_s), _lib.ptr(out_i), _lib.ptr(out_c), _lib.ptr(ws),
                    ws_bytes, _lib.current_stream(),
                ),
                "rp_sim_topk_fp8",
            )
            return out_i, out_s, out_c
        _lib.check(
            lib.rp_sim_topk(
                _lib.ptr(Q), _lib.ptr(E), B, N, D, _lib.ptr(file_of), _lib.ptr(end_key), _lib.ptr(d_bits),
                len(self._files), _lib.ptr(d_own), _lib.ptr(d_qk), 0, k, flags, _lib.ptr(out_s), _lib.ptr(out_i),
                _lib.ptr(out_c), _lib.ptr(ws), ws_bytes, _lib.current_stream(),
            ),
            "rp_sim_topk",
        )
        return out_i, out_s, out_c

    def _nearest_premise_ids_paged(self, premise_embeddings, batch_context, batch_context_emb, k: int, dense: bool):
        lib = _lib.load()
        dev = batch_context_emb.device
        if dev.type != "cuda":
            raise _lib.HipLibraryError("nearest-premise search runs on the GPU only (no CPU fallback)")
        fp8 = isinstance(premise_embeddings, Fp8Index)
        E = premise_embeddings if fp8 else as_bf16_matrix(premise_embeddings, dev)
        Q = Fp8Index.quantize(batch_context_emb) if fp8 else as_bf16_matrix(batch_context_emb, dev)
        B, D = Q.shape
        N = E.shape[0]
        file_of, end_key = self._device_arrays(dev)
        d_bits, d_own, d_qk = self.device_query_masks(batch_context, dev)
        out_s = torch.full((B, k), float("-inf"), dtype=torch.float32, device=dev)
        out_i = torch.full((B, k), -1, dtype=torch.int32, device=dev)
        out_c = torch.zeros((B,), dtype=torch.int32, device=dev)
        flags = _lib.RP_TOPK_DENSE if dense else _lib.RP_TOPK_AUTO
        P = MAX_K_PER_CALL
        ws_bytes = lib.rp_sim_topk_workspace_bytes(B, N, D, P, flags)
        ws = _workspace(dev, ws_bytes)
        pg_s = torch.empty((B, P), dtype=torch.float32, device=dev)
        pg_i = torch.empty((B, P), dtype=torch.int32, device=dev)
        pg_c = torch.empty((B,), dtype=torch.int32, device=dev)
        after_s = torch.zeros((B,), dtype=torch.float32, device=dev)
        after_i = torch.full((B,), -1, dtype=torch.int32, device=dev)
        masks = (_lib.ptr(file_of), _lib.ptr(end_key), _lib.ptr(d_bits), len(self._files), _lib.ptr(d_own), _lib.ptr(d_qk), 0)
        for lo in range(0, k, P):
            kk = min(P, k - lo)
            # Every page is a call with the FULL page size P (the last one is truncated here, on the host): the plan
            # (dense / two-pass, sampling stride, tile configurations) depends on k, so equal k means every page scores a
            # row with the same kernels - "pages concatenate exactly" then never rests on two kernels agreeing to the bit
            # at a page boundary.
            outs = (P, flags, _lib.ptr(pg_s), _lib.ptr(pg_i), _lib.ptr(pg_c), _lib.ptr(ws), ws_bytes, _lib.current_stream())
            if fp8:
                head = (_lib.ptr(Q.codes), _lib.ptr(Q.scale), _lib.ptr(E.codes), _lib.ptr(E.scale), B, N, D)
                st = (lib.rp_sim_topk_fp8(*head, *masks, *outs) if lo == 0 else
                      lib.rp_sim_topk_fp8_after(*head, *masks, _lib.ptr(after_s), _lib.ptr(after_i), *outs))
            else:
                head = (_lib.ptr(Q), _lib.ptr(E), B, N, D)
                st = (lib.rp_sim_topk(*head, *masks, *outs) if lo == 0 else
                      lib.rp_sim_topk_after(*head, *masks, _lib.ptr(after_s), _lib.ptr(after_i), *outs))
            _lib.check(st, "rp_sim_topk (page)")
            if bool((pg_c < 0).any()):  # candidate overflow on this page: the whole search again, dense plan
                if dense:
                    raise _lib.HipLibraryError("rp_sim_topk reported an overflow under RP_TOPK_DENSE")
                return self._nearest_premise_ids_paged(premise_embeddings, batch_context, batch_context_emb, k, True)
            out_s[:, lo : lo + kk] = pg_s[:, :kk]
            out_i[:, lo : lo + kk] = pg_i[:, :kk]
            out_c += pg_c.clamp(max=kk)
            # the next page starts behind this page's last entry; a query whose page came back short has nothing left
            full = pg_c == P
            after_s = torch.where(full, pg_s[:, P - 1], torch.full_like(after_s, float("-inf"))).contiguous()
            after_i = torch.where(full, pg_i[:, P - 1], torch.full_like(after_i, 2 ** 31 - 1)).contiguous()
            if not bool(full.any()):
                break
        return out_i, out_s, out_c

    def launch_nearest_premises(
        self,
        premise_embeddings: "torch.Tensor | Fp8Index",
        batch_context: Sequence[Context],
        batch_context_emb: torch.Tensor,
        k: int,
        also_copy: Sequence[torch.Tensor] = (),
    ) -> "PendingSearch":
        """First half of ``get_nearest_premises``: everything up to and including the asynchronous copy of the result
        to pinned host memory is enqueued; nothing waits.  ``finish()`` of the returned object is the second half.
        ``also_copy``: small device tensors (the encoder's verdict words) to bring along: ``.extra_host`` after finish."""
        ids, scores, counts = self.nearest_premise_ids(premise_embeddings, batch_context, batch_context_emb, k)
        B = len(batch_context)
        host = _pinned_result_buffers(B, k)
        host[0].copy_(ids, non_blocking=True)
        host[1].copy_(scores, non_blocking=True)
        host[2].copy_(counts, non_blocking=True)
        extra = [_pinned_small(t).copy_(t, non_blocking=True) for t in also_copy]
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(batch_context_emb.device))
        return PendingSearch(self, premise_embeddings, list(batch_context), batch_context_emb, k, host, done, extra)

    def get_nearest_premises(
        self,
        premise_embeddings: torch.Tensor,
        batch_context: List[Context],
        batch_context_emb: torch.Tensor,
        k: int,
    ) -> Tuple[List[List[Premise]], List[List[float]]]:
        """Batch nearest-neighbour search restricted to accessible premises (common.py:299-326).
        Raises ``ValueError`` when a query has fewer than ``k`` accessible premises, as the
        reference does (common.py:323-324)."""
        return self.launch_nearest_premises(premise_embeddings, batch_context, batch_context_emb, k).finish()


_STAGING_SLOTS = 4  # pinned staging buffers used in turn (a slot is reused four uploads later: long completed)
_staging: List[list] = []  # [[pinned uint8 buffer, event of its last upload], ...]; buffers grow to the largest batch seen
_staging_next = 0


def _upload_query_keys(own: np.ndarray, qk: np.ndarray, dev: torch.device):
    """(own_file int32 [B], q_key int64 [B]) on the device through ONE asynchronous copy from pinned staging memory
    (``.to(device)`` from pageable memory would block the host until the stream reaches the copy - i.e. until the
    encode launched just before has finished).  The staging ring is bounded: ``_STAGING_SLOTS`` buffers in all, each as
    large as the largest batch seen."""
    global _staging_next
    B = own.shape[0]
    n_qk, n_own = 8 * B, 4 * B
    total = n_qk + n_own
    if len(_staging) < _STAGING_SLOTS:
        _staging.append([torch.empty(max(total, 4096), dtype=torch.uint8).pin_memory(), None])
        slot = _staging[-1]
    else:
        slot = _staging[_staging_next]
        _staging_next = (_staging_next + 1) % _STAGING_SLOTS
        if slot[1] is not None:
            slot[1].synchronize()
        if slot[0].numel() < total:
            slot[0] = torch.empty(total, dtype=torch.uint8).pin_memory()
    h = slot[0].numpy()
    h[:n_qk].view(np.int64)[:] = qk
    h[n_qk:total].view(np.int32)[:] = own
    d = torch.empty(total, dtype=torch.uint8, device=dev)
    d.copy_(slot[0][:total], non_blocking=True)
    slot[1] = torch.cuda.Event()
    slot[1].record(torch.cuda.current_stream(dev))
    return d[n_qk:].view(torch.int32), d[:n_qk].view(torch.int64)


_small_ring: List[torch.Tensor] = []
_small_next = 0


def _pinned_small(like: torch.Tensor) -> torch.Tensor:
    """A pinned host tensor for a few words (the encoder's verdict) from a ring of 16 preallocated 64-byte slots:
    ``pin_memory()`` is a slow host call and this sits on every predict_step."""
    global _small_next
    nbytes = like.numel() * like.element_size()
    if nbytes > 64:
        return torch.empty(like.shape, dtype=like.dtype).pin_memory()
    if not _small_ring:
        base = torch.empty(16 * 64, dtype=torch.uint8).pin_memory()
        _small_ring.extend(base[i * 64 : (i + 1) * 64] for i in range(16))
    slot = _small_ring[_small_next]
    _small_next = (_small_next + 1) % 16
    return slot[:nbytes].view(like.dtype).view(like.shape)


_PINNED_POOL_SHAPES = 8  # distinct (B, k) result shapes kept (least recently used dropped: pinned memory is bounded)
_pinned_pool: Dict[Tuple[int, int], List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]] = {}


def _pinned_result_buffers(B: int, k: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(ids int32 [B,k], scores f32 [B,k], counts int32 [B]) in pinned host memory, from a small pool per shape
    (page-locking costs more than the copy); a set goes back to the pool when its search has been finished."""
    pool = _pinned_pool.pop((B, k), [])
    _pinned_pool[(B, k)] = pool  # most recently used last
    while len(_pinned_pool) > _PINNED_POOL_SHAPES:
        _pinned_pool.pop(next(iter(_pinned_pool)))
    if pool:
        return pool.pop()
    return (torch.empty((B, k), dtype=torch.int32).pin_memory(), torch.empty((B, k), dtype=torch.float32).pin_memory(),
            torch.empty((B,), dtype=torch.int32).pin_memory())


class PendingSearch:
    """A launched nearest-premise search whose result is on its way to the host (``Corpus.launch_nearest_premises``)."""

    def __init__(self, corpus, premise_embeddings, batch_context, batch_context_emb, k, host, done, extra_host=()):
        self.corpus, self.premise_embeddings, self.batch_context = corpus, premise_embeddings, batch_context
        self.batch_context_emb, self.k, self.host, self.done = batch_context_emb, k, host, done
        self.extra_host = list(extra_host)

    def finish(self) -> Tuple[List[List[Premise]], List[List[float]]]:
        """Wait for the copy, map ids to ``Premise`` objects; ``ValueError`` as the reference (common.py:323-324)."""
        self.done.synchronize()
        ids_h, scores_h, counts_h = self.host
        k = self.k
        try:
            if bool((counts_h < 0).any()):  # candidate-list overflow (reserved by the ABI): redo with the dense pass
                ids, scores, counts = self.corpus.nearest_premise_ids(
                    self.premise_embeddings, self.batch_context, self.batch_context_emb, k, dense=True
                )
                ids_h, scores_h, counts_h = ids.cpu(), scores.cpu(), counts.cpu()
            if bool((counts_h < k).any()):
                raise ValueError
            ids_l, scores_l = ids_h.tolist(), scores_h.tolist()
        finally:
            pool = _pinned_pool.get((len(self.batch_context), k))
            if pool is not None and len(pool) < 4:
                pool.append(self.host)
            self.batch_context_emb = None
        prem = self.corpus.all_premises
        return [[prem[i] for i in row] for row in ids_l], scores_l


@dataclass
class Fp8Index:
    """An embedding matrix in OCP e4m3 with one fp32 scale per row (BASELINE.json configs[4]; no
    reference counterpart — the reference keeps the index in the model dtype).  Row i dequantises
    to ``codes[i].float8_e4m3fn * scale[i]``; half the bytes of the bf16 index for the scan to read.
    Accepted wherever ``premise_embeddings`` is: ``Corpus.get_nearest_premises(Fp8Index, ...)``."""

    codes: torch.Tensor  # uint8 [N, D], device
    scale: torch.Tensor  # float32 [N], device

    @classmethod
    def quantize(cls, embeddings: torch.Tensor, device: Optional[torch.device] = None) -> "Fp8Index":
        """``rp_quantize_rows_e4m3`` over a [N, D] fp32 / bf16 matrix (moved to ``device`` first)."""
        lib = _lib.load()
        dev = torch.device(device) if device is not None else embeddings.device
        if dev.type != "cuda":
            raise _lib.HipLibraryError("e4m3 quantisation runs on the GPU only (no CPU fallback)")
        X = embeddings.to(dev)
        if X.dtype not in (torch.float32, torch.bfloat16):
            X = X.float()
        X = X.contiguous()
        N, D = X.shape
        codes = torch.empty((N, D), dtype=torch.uint8, device=dev)
        scale = torch.empty((N,), dtype=torch.float32, device=dev)
        dt = _lib.RP_DT_F32 if X.dtype == torch.float32 else _lib.RP_DT_BF16
        _lib.check(lib.rp_quantize_rows_e4m3(_lib.ptr(X), dt, N, D, _lib.ptr(codes), _lib.ptr(scale),
                                             _lib.current_stream()), "rp_quantize_rows_e4m3")
        return cls(codes, scale)

    @property
    def shape(self) -> Tuple[int, int]:
        return tuple(self.codes.shape)

    @property
    def device(self) -> torch.device:
        return self.codes.device

    def __len__(self) -> int:
        return self.codes.shape[0]

    def dequantize(self) -> torch.Tensor:
        """fp32 [N, D] (checks and tests; the scan never materialises this)."""
        return self.codes.view(torch.float8_e4m3fn).float() * self.scale[:, None]


_ws_cache: Dict[str, torch.Tensor] = {}


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """A grow-only scratch buffer per device (the C ABI never allocates after create)."""
    key = str(device)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


_bf16_cache: List[Tuple[torch.Tensor, str, torch.Tensor]] = []  # at most one (source tensor, device, bf16 copy)


def drop_cast_cache() -> None:
    """Forget the cached bf16 cast (call after rewriting a cached source matrix in place)."""
    _bf16_cache.clear()


def as_bf16_matrix(t: torch.Tensor, device: torch.device) -> torch.Tensor:
    """Contiguous bf16 device view/copy of an embedding matrix; one large conversion is cached so that
    ``retrieve`` does not re-cast the corpus every call (the reference caches the cast in
    ``self.corpus_embeddings``, model.py:363-366).  The cache is keyed on the source tensor OBJECT, held
    by a strong reference together with its ``_version``: a freed-and-recycled storage can therefore
    never alias the key, and in-place torch writes invalidate it; writers that go through raw pointers
    call ``drop_cast_cache``."""
    if t.dtype == torch.bfloat16 and t.device == device and t.is_contiguous():
        return t
    if t.numel() < (1 << 20):
        return t.to(device=device, dtype=torch.bfloat16).contiguous()
    for src, dev_key, hit in _bf16_cache:
        if src is t and dev_key == f"{device}:{t._version}":
            return hit
    _bf16_cache.clear()
    hit = t.to(device=device, dtype=torch.bfloat16).contiguous()
    _bf16_cache.append((t, f"{device}:{t._version}", hit))
    return hit


@dataclass(frozen=True)
class IndexedCorpus:
    """A corpus with its premise embeddings (common.py:329-338)."""

    corpus: Corpus
    embeddings: torch.Tensor

    def __post_init__(self):
        assert self.embeddings.device == torch.device("cpu")
        assert len(self.embeddings) == len(self.corpus)


# ----------------------------------------------------------------------------------------------
# Native on-disk index.  The reference persists ``pickle(IndexedCorpus(corpus,
# fp32 embeddings))`` (retrieval/index.py:37-40): 765 MB of fp32 plus a pickled object graph that
# drags in networkx and lean_dojo class identities.  The native form is a directory:
#     corpus.jsonl            the corpus exactly as given (App. B.1), so every field survives
#     embeddings.safetensors  "embeddings": [N, D] in the encoder's dtype (bf16: 383 MB at 130k x 1472)
#     arrays.safetensors      what the GPU search consumes, precomputed: "reach" int64 [F, ceil(F/64)] (transitive
#                             import closure as bit rows), "file_of" int32 [N], "end_key" int64 [N],
#                             "file_start" int64 [F+1]  - loading never re-runs the closure
#     fp8.safetensors         optional e4m3 form of the index: "codes" uint8 [N, D], "scale" f32 [N]
#     meta.json               {"format": 2, "n_premises": N, "n_files": F, "d_model": D, "dtype": "...", "fp8": bool}
# ``PremiseRetriever.load_corpus`` accepts a ``.jsonl``, a pickle, or such a directory.
# ----------------------------------------------------------------------------------------------
INDEX_FORMAT_VERSION = 2


def save_index(dir_path: str, corpus_jsonl_path: str, embeddings: torch.Tensor, corpus: Optional[Corpus] = None,
               fp8: Optional["Fp8Index"] = None) -> None:
    import os
    import shutil

    from safetensors.torch import save_file

    os.makedirs(dir_path, exist_ok=True)
    dst = os.path.join(dir_path, "corpus.jsonl")
    if os.path.abspath(corpus_jsonl_path) != os.path.abspath(dst):
        shutil.copyfile(corpus_jsonl_path, dst)
    if corpus is None:
        corpus = Corpus(dst)
    emb = embeddings.detach().cpu().contiguous()
    assert emb.shape[0] == len(corpus)
    save_file({"embeddings": emb}, os.path.join(dir_path, "embeddings.safetensors"))
    save_file({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in corpus.index_arrays().items()},
              os.path.join(dir_path, "arrays.safetensors"))
    if fp8 is not None:
        assert len(fp8) == len(corpus)
        save_file({"codes": fp8.codes.detach().cpu().contiguous(), "scale": fp8.scale.detach().cpu().contiguous()},
                  os.path.join(dir_path, "fp8.safetensors"))
    with open(os.path.join(dir_path, "meta.json"), "w") as fh:
        json.dump({"format": INDEX_FORMAT_VERSION, "n_premises": int(emb.shape[0]), "n_files": corpus.num_files,
                   "d_model": int(emb.shape[1]), "dtype": str(emb.dtype).replace("torch.", ""), "fp8": fp8 is not None},
                  fh)


def load_index(dir_path: str, with_fp8: bool = False):
    """(corpus, embeddings) - or (corpus, embeddings, (codes, scale) | None) with ``with_fp8`` - from a native index
    directory.  Format 2 carries the closure bit rows and the per-premise arrays, so nothing is rebuilt; a format-1
    directory (round 1: embeddings + jsonl only) is still accepted and rebuilds them."""
    import os

    from safetensors.torch import load_file

    with open(os.path.join(dir_path, "meta.json")) as fh:
        meta = json.load(fh)
    if meta.get("format") not in (1, INDEX_FORMAT_VERSION):
        raise ValueError(f"unsupported index format {meta.get('format')!r} in {dir_path}")
    arrays = None
    if meta["format"] >= 2:
        arrays = {k: v.numpy() for k, v in load_file(os.path.join(dir_path, "arrays.safetensors")).items()}
    corpus = Corpus(os.path.join(dir_path, "corpus.jsonl"), _arrays=arrays)
    emb = load_file(os.path.join(dir_path, "embeddings.safetensors"))["embeddings"]
    assert emb.shape == (meta["n_premises"], meta["d_model"]) and len(corpus) == meta["n_premises"]
    if not with_fp8:
        return corpus, emb
    payload = None
    if meta.get("fp8"):
        t = load_file(os.path.join(dir_path, "fp8.safetensors"))
        payload = (t["codes"], t["scale"])
    return corpus, emb, payload


# ----------------------------------------------------------------------------------------------------------------
# The reference's own index file.  `python retrieval/index.py` (retrieval/index.py:37-40) pickles
# ``IndexedCorpus(corpus, embeddings)`` with the REFERENCE's classes inside: ``common.Corpus`` (a networkx DiGraph
# of ``common.File`` nodes), ``common.Premise``, ``lean_dojo.Pos`` - the file a prover user actually has
# (prover/tactic_generator.py:273-276 loads it).  None of those modules exist here, so the stream is read with a
# class map: every foreign class becomes a plain attribute bag, and the corpus is rebuilt from what the bags hold -
# the files in the graph's node order, their premises, and the closure's successor lists as imports.
# ----------------------------------------------------------------------------------------------------------------
class _Bag:
    """Stand-in for a class of the reference stream: keeps whatever state pickle hands it."""

    def __setstate__(self, state):
        if isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):  # (dict state, slots state)
            self.__dict__.update(state[0] or {})
            self.__dict__.update(state[1])
        elif isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["_state"] = state


class _ReferenceUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        root = module.split(".")[0]
        if root == "reprover_amd" or root in ("torch", "numpy", "collections", "builtins", "copyreg", "_codecs"):
            return super().find_class(module, name)
        if root in ("common", "lean_dojo", "networkx", "retrieval"):
            return type(name, (_Bag,), {"_foreign": f"{module}.{name}"})
        return super().find_class(module, name)


def _is_foreign(obj, name: str) -> bool:
    return getattr(type(obj), "_foreign", "").endswith("." + name)


def _pos_of(bag) -> Pos:
    return bag if isinstance(bag, Pos) else Pos(int(bag.line_nb), int(bag.column_nb))


def load_indexed_corpus_pickle(path: str):
    """(Corpus, embeddings) from a pickled ``IndexedCorpus`` - this package's own or the reference's
    (retrieval/index.py:37-40; read without ``lean_dojo`` / ``networkx`` / the reference's ``common`` module)."""
    with open(path, "rb") as fh:
        obj = _ReferenceUnpickler(fh).load()
    if isinstance(obj, IndexedCorpus):
        return obj.corpus, obj.embeddings
    if not _is_foreign(obj, "IndexedCorpus"):
        raise TypeError(f"{path} holds a {type(obj).__name__}, not an IndexedCorpus")
    graph = obj.corpus.transitive_dep_graph
    nodes, succ = graph._node, getattr(graph, "_succ", None) or graph._adj
    files = []
    for fpath, attrs in nodes.items():  # node order = corpus.jsonl order (common.py:199-213): topological
        fb = attrs["file"]
        prem = [Premise(p.path, p.full_name, _pos_of(p.start), _pos_of(p.end), p.code) for p in fb.premises]
        files.append((File(fb.path, prem), list(succ[fpath])))
    corpus = Corpus.from_files(files)
    if len(corpus) != len(obj.corpus.all_premises):
        raise ValueError("the pickled corpus' premise list does not match its files")
    return corpus, obj.embeddings


# ----------------------------------------------------------------------------------------------------------------
# ... and the other direction: an index file the REFERENCE loads.  Its prover unpickles `common.IndexedCorpus`
# (prover/tactic_generator.py:273-276 -> retrieval/model.py:81-85), so the stream must name the reference's own
# classes - `common.{IndexedCorpus, Corpus, File, Premise}`, `lean_dojo.data_extraction.lean.Pos` - and hold what their
# instances hold: `Corpus.{all_premises, transitive_dep_graph (a networkx DiGraph of the closure, node attribute
# "file"), imported_premises_cache}` (common.py:181-224, 264-278).  The objects below are attribute-for-attribute
# stand-ins pickled under those names (plain `object.__reduce_ex__` state, which is also how the reference's
# dataclasses pickle); while the stream is written the names resolve to the stand-ins, afterwards `sys.modules` is as
# it was.  tests/golden/make_golden.py (g17) loads such a file with the imported reference's plain `pickle.load` and runs
# its `retrieve` on it.
# ----------------------------------------------------------------------------------------------------------------
REFERENCE_POS_MODULE = "lean_dojo.data_extraction.lean"  # where lean_dojo defines Pos (`from lean_dojo import Pos`)


def _reference_standins() -> Dict[str, type]:
    def make(module, name):
        return type(name, (), {"__module__": module, "__qualname__": name})

    return {"IndexedCorpus": make("common", "IndexedCorpus"), "Corpus": make("common", "Corpus"),
            "File": make("common", "File"), "Premise": make("common", "Premise"), "Pos": make(REFERENCE_POS_MODULE, "Pos")}


def save_reference_pickle(path: str, corpus: "Corpus", embeddings: torch.Tensor) -> None:
    """Write ``IndexedCorpus(corpus, fp32 CPU embeddings)`` as the reference's OWN pickle (retrieval/index.py:37-40):
    its ``pickle.load`` - with ``common``, ``lean_dojo`` and ``networkx`` importable, as in the reference's environment -
    returns a ``common.IndexedCorpus`` that ``PremiseRetriever.load_corpus`` / the prover use as they use their own."""
    import sys
    import types

    try:
        import networkx as nx
    except ImportError as e:  # the reference's Corpus IS a networkx graph; without the package its state cannot be written
        raise ImportError("save_reference_pickle needs networkx (the reference's Corpus holds a networkx.DiGraph)") from e
    emb = embeddings.detach().to(torch.float32).cpu().contiguous()
    if len(emb) != len(corpus):
        raise ValueError(f"{len(emb)} embedding rows for {len(corpus)} premises")
    cls = _reference_standins()

    def obj(kind, **state):
        o = cls[kind].__new__(cls[kind])
        o.__dict__.update(state)
        return o

    def pos(p):
        return obj("Pos", line_nb=int(p.line_nb), column_nb=int(p.column_nb))

    graph = nx.DiGraph()
    prem_of: Dict[int, Any] = {}
    files = []
    for f in corpus._files:
        prem = []
        for p in f.premises:
            q = prem_of[id(p)] = obj("Premise", path=p.path, full_name=p.full_name, start=pos(p.start), end=pos(p.end), code=p.code)
            prem.append(q)
        files.append(obj("File", path=f.path, premises=prem))
        graph.add_node(f.path, file=files[-1])
    for i, f in enumerate(corpus._files):  # the graph IS the transitive closure (common.py:216)
        for g in corpus._reach_ids(i):
            graph.add_edge(f.path, corpus._files[int(g)].path)
    cache = {f.path: [q for g in corpus._reach_ids(i) for q in files[int(g)].premises] for i, f in enumerate(corpus._files)}
    ref_corpus = obj("Corpus", all_premises=[prem_of[id(p)] for p in corpus.all_premises], transitive_dep_graph=graph,
                     imported_premises_cache=cache)
    indexed = obj("IndexedCorpus", corpus=ref_corpus, embeddings=emb)

    names = ["common", "lean_dojo", "lean_dojo.data_extraction", REFERENCE_POS_MODULE]
    saved = {n: sys.modules.get(n) for n in names}
    try:
        common_mod = types.ModuleType("common")
        for n in ("IndexedCorpus", "Corpus", "File", "Premise"):
            setattr(common_mod, n, cls[n])
        sys.modules["common"] = common_mod
        for n in names[1:]:
            sys.modules[n] = types.ModuleType(n)
        sys.modules[REFERENCE_POS_MODULE].Pos = cls["Pos"]
        with open(path, "wb") as fh:
            pickle.dump(indexed, fh, protocol=4)
    finally:
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m


def get_all_pos_premises(annot_tac, corpus: Corpus) -> List[Premise]:
    """Premises used by an annotated tactic: each provenance ``{def_path, def_pos}`` is resolved with
    ``corpus.locate_premise``; unresolvable ones are skipped (common.py:341-354)."""
    _, provenances = annot_tac
    found = set()
    for prov in provenances:
        p = corpus.locate_premise(prov["def_path"], Pos(*prov["def_pos"]))
        if p is not None:
            found.add(p)
    return list(found)


def format_augmented_state(s: str, premises: List[Premise], max_len: Optional[int] = None, p_drop: float = 0.0) -> str:
    """Byte-budgeted concatenation of retrieved premises in front of a state (common.py:357-378);
    the caller-side consumer of ``retrieve`` (prover/tactic_generator.py:293-295)."""
    import random

    budget = (max_len if max_len is not None else 9999999999999999999999) - len(s.encode("utf-8"))
    aug, used = "", 0
    for p in premises:
        if random.random() < p_drop:
            continue
        piece = f"{p.serialize()}\n\n"
        n = len(piece.encode("utf-8"))
        if used + n > budget:
            continue
        used += n
        aug = piece + aug
    return aug + s


def zip_strict(*args):
    """common.py:428-431."""
    assert len(args) > 1 and all(len(args[0]) == len(a) for a in args[1:])
    return zip(*args)
