"""Oracle (test infrastructure only): the corpus data model and nearest-premise search.

Plain-Python / numpy restatement of the reference's ``common.py`` pieces that sit on the
retrieval hot path.  Each function cites the reference lines it follows
(paths relative to /root/reference).  Slow and literal on purpose.
"""
from __future__ import annotations

import json
import re
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

MARK_START = "<a>"  # common.py:25
MARK_END = "</a>"  # common.py:26


class Pos(tuple):
    """``lean_dojo.Pos`` value type as the reference uses it:
    ``Pos(line_nb, column_nb)``, iterable, totally ordered lexicographically
    (used with ``<=`` at common.py:89, 260, 286, 295)."""

    def __new__(cls, line_nb: int, column_nb: int):
        return super().__new__(cls, (int(line_nb), int(column_nb)))

    @property
    def line_nb(self) -> int:
        return self[0]

    @property
    def column_nb(self) -> int:
        return self[1]


class ContextRef:
    """common.py:34-56.  Equality/hash ignore ``theorem_pos`` (field(compare=False))."""

    def __init__(self, path: str, theorem_full_name: str, theorem_pos: Pos, state: str):
        assert isinstance(path, str) and isinstance(theorem_full_name, str)
        assert isinstance(theorem_pos, Pos)
        assert isinstance(state, str) and "⊢" in state  # common.py:46-52
        assert MARK_START not in state and MARK_END not in state
        self.path, self.theorem_full_name, self.theorem_pos, self.state = (
            path,
            theorem_full_name,
            theorem_pos,
            state,
        )

    def serialize(self) -> str:  # common.py:54-56
        return self.state


class PremiseRef:
    """common.py:59-106.  Equality/hash use (path, full_name, start) only."""

    def __init__(self, path: str, full_name: str, start: Pos, end: Pos, code: str):
        assert isinstance(path, str) and isinstance(full_name, str)
        assert isinstance(start, Pos) and isinstance(end, Pos) and start <= end  # :86-90
        assert isinstance(code, str) and code != ""  # :91
        self.path, self.full_name, self.start, self.end, self.code = path, full_name, start, end, code

    def _key(self):
        return (self.path, self.full_name, self.start)

    def __eq__(self, other):
        return isinstance(other, PremiseRef) and self._key() == other._key()

    def __hash__(self):
        return hash(self._key())

    def serialize(self) -> str:
        """common.py:93-106: mark the premise's own name inside its code with <a>…</a>.

        First every literal ``_root_.<full_name>`` is replaced; then, for the dotted suffixes
        of the name from longest to shortest, the first suffix whose regex substitution
        (lookbehind for whitespace, optional «» guillemets, *unescaped* suffix used as the
        pattern) changes the code wins.
        """
        marked = f"{MARK_START}{self.full_name}{MARK_END}"
        code = self.code.replace(f"_root_.{self.full_name}", marked)
        parts = self.full_name.split(".")
        for i in range(len(parts)):
            suffix = ".".join(parts[i:])
            replaced = re.sub("(?<=\\s)«?" + suffix + "»?", marked, code)
            if replaced != code:
                return replaced
        return code


def premises_of_file(file_data: dict) -> List[PremiseRef]:
    """common.py:153-173 (File.from_data): which premise records survive."""
    out = []
    for rec in file_data["premises"]:
        name = rec["full_name"]
        if name is None:  # :160-161
            continue
        if "user__.n" in name or rec["code"] == "":  # :162-164
            continue
        if name.startswith("[") and name.endswith("]"):  # :165-167
            continue
        out.append(PremiseRef(file_data["path"], name, Pos(*rec["start"]), Pos(*rec["end"]), rec["code"]))
    return out


class CorpusRef:
    """common.py:181-326: a DAG of files, premises in file order, transitive imports."""

    def __init__(self, jsonl_path: str):
        self.paths: List[str] = []  # insertion (= topological) order, common.py:200-213
        self.file_premises: Dict[str, List[PremiseRef]] = {}
        self.direct: Dict[str, List[str]] = {}
        self.all_premises: List[PremiseRef] = []
        with open(jsonl_path) as fh:
            for line in fh:
                data = json.loads(line)
                path = data["path"]
                assert path not in self.file_premises  # :204
                prem = premises_of_file(data)
                self.file_premises[path] = prem
                self.paths.append(path)
                self.all_premises.extend(prem)  # :208
                for imp in data["imports"]:
                    assert imp in self.file_premises  # :211 (imports precede importers)
                self.direct[path] = list(data["imports"])
        # Transitive closure (reference: networkx transitive_closure_dag, :215).  Files arrive in
        # topological order, so one forward sweep suffices.
        self.reach: Dict[str, set] = {}
        for path in self.paths:
            r = set()
            for imp in self.direct[path]:
                r.add(imp)
                r |= self.reach[imp]
            self.reach[path] = r

    def __len__(self) -> int:  # :223
        return len(self.all_premises)

    def get_premises(self, path: str) -> List[PremiseRef]:  # :245-247
        return self.file_premises[path]

    def locate_premise(self, path: str, pos: Pos) -> Optional[PremiseRef]:  # :253-262
        for p in self.get_premises(path):
            if p.start <= pos <= p.end:
                return p
        return None

    def accessible_keys(self, path: str, pos: Pos) -> set:
        """common.py:280-289 + PremiseSet (:109-138): the accessible set, as the
        ``(path, full_name)`` keys PremiseSet indexes by."""
        keys = set()
        for p in self.get_premises(path):
            if p.end <= pos:  # :286
                keys.add((p.path, p.full_name))
        for dep in self.reach[path]:  # :288 via _get_imported_premises :268-278
            for p in self.file_premises[dep]:
                keys.add((p.path, p.full_name))
        return keys

    def accessible_indexes(self, path: str, pos: Pos) -> List[int]:
        """common.py:291-297 (get_accessible_premise_indexes)."""
        reach = self.reach[path]
        return [
            i
            for i, p in enumerate(self.all_premises)
            if (p.path == path and p.end <= pos) or (p.path in reach)
        ]

    def get_nearest_premises(
        self,
        premise_embeddings: np.ndarray,
        batch_context: Sequence[ContextRef],
        batch_context_emb: np.ndarray,
        k: int,
    ) -> Tuple[List[List[int]], List[List[float]]]:
        """common.py:299-326.  Returns premise *indexes* into ``all_premises`` and scores.

        similarities = Q @ E.T (:307); per query walk the ids in descending-similarity order
        (:308) keeping accessible premises until k are found (:312-322); ValueError when the
        accessible set is exhausted first (:323-324).  The reference's ``argsort`` leaves the
        order of exact ties unspecified; the oracle fixes it: equal scores → lower index first.
        """
        E = np.asarray(premise_embeddings, dtype=np.float32)
        Q = np.asarray(batch_context_emb, dtype=np.float32)
        sims = Q @ E.T
        out_idx: List[List[int]] = []
        out_score: List[List[float]] = []
        for j, ctx in enumerate(batch_context):
            order = np.argsort(-sims[j], kind="stable")
            keys = self.accessible_keys(ctx.path, ctx.theorem_pos)
            got_i: List[int] = []
            got_s: List[float] = []
            for i in order:
                p = self.all_premises[int(i)]
                if (p.path, p.full_name) in keys:
                    got_i.append(int(i))
                    got_s.append(float(sims[j, i]))
                    if len(got_i) >= k:
                        break
            else:
                raise ValueError("fewer than k accessible premises")  # :323-324
            out_idx.append(got_i)
            out_score.append(got_s)
        return out_idx, out_score


def masked_topk(sims: np.ndarray, accessible: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Array form of common.py:308-324 for checking kernels on raw matrices: for each row of
    ``sims`` [B,N], the k best columns among ``accessible`` [B,N] (bool), ordered by
    (score descending, index ascending).  Raises ValueError if a row has < k accessible."""
    B, N = sims.shape
    ids = np.zeros((B, k), dtype=np.int64)
    sc = np.zeros((B, k), dtype=np.float32)
    for j in range(B):
        cand = np.flatnonzero(accessible[j])
        if cand.size < k:
            raise ValueError("fewer than k accessible premises")
        order = cand[np.argsort(-sims[j, cand], kind="stable")][:k]
        ids[j] = order
        sc[j] = sims[j, order]
    return ids, sc


def format_augmented_state(s: str, premise_texts: Iterable[str], max_len: Optional[int], p_drop: float = 0.0) -> str:
    """common.py:357-378: prepend serialized premises (each followed by a blank line) while their
    total UTF-8 byte length fits in ``max_len - len(bytes(s))``; later premises end up *earlier* in
    the string; one ``random.random() < p_drop`` draw per premise, in list order, BEFORE the size test."""
    import random

    budget = (max_len if max_len is not None else 9999999999999999999999) - len(s.encode("utf-8"))
    aug, used = "", 0
    for text in premise_texts:
        if random.random() < p_drop:
            continue
        piece = f"{text}\n\n"
        n = len(piece.encode("utf-8"))
        if used + n > budget:
            continue
        used += n
        aug = piece + aug
    return aug + s
