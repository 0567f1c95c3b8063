"""ByT5 byte tokenizer, vectorised.

Stands in for ``AutoTokenizer.from_pretrained(<byt5>)`` on the retrieval path
(retrieval/model.py:44, 199-205, 351-357; retrieval/datamodule.py:135-141): UTF-8 bytes + 3,
EOS = 1 appended, truncation to ``max_length`` (the last kept token is always EOS), right
padding with 0.  HuggingFace's slow tokenizer walks the string one Python ``chr`` at a time
(tokenization_byt5.py:195-208); here a string is one ``np.frombuffer``.  Pinned against HF by
tests/golden/g1_tokenizer.json.
"""
from __future__ import annotations

import re
from typing import List, Sequence, Tuple

import numpy as np
import torch

PAD_TOKEN_ID, EOS_TOKEN_ID, UNK_TOKEN_ID = 0, 1, 2
_OFFSET = 3
_EXTRA_BASE = 259
_N_EXTRA = 125
# pad/eos/unk are AddedTokens with lstrip=rstrip=True: they swallow surrounding whitespace.
_SPECIALS = re.compile(r"\s*(</s>|<unk>|<pad>)\s*|<extra_id_(\d+)>")
_SPECIAL_IDS = {"<pad>": PAD_TOKEN_ID, "</s>": EOS_TOKEN_ID, "<unk>": UNK_TOKEN_ID}


def _bytes_to_ids(b: bytes) -> np.ndarray:
    return np.frombuffer(b, dtype=np.uint8).astype(np.int32) + _OFFSET


def encode_one(text: str, max_length: int) -> np.ndarray:
    """ids (int32) of one string including the final EOS, truncated to ``max_length``."""
    if "<" not in text:  # no special token can occur: the common case for Lean source
        body = _bytes_to_ids(text.encode("utf-8")[: max_length - 1])
        return np.concatenate([body, np.array([EOS_TOKEN_ID], dtype=np.int32)])
    parts: List[np.ndarray] = []
    pos = 0
    for m in _SPECIALS.finditer(text):
        if m.group(1) is None and int(m.group(2)) >= _N_EXTRA:
            continue  # not in the vocabulary: plain bytes
        parts.append(_bytes_to_ids(text[pos : m.start()].encode("utf-8")))
        tok = _SPECIAL_IDS[m.group(1)] if m.group(1) is not None else _EXTRA_BASE + int(m.group(2))
        parts.append(np.array([tok], dtype=np.int32))
        pos = m.end()
    parts.append(_bytes_to_ids(text[pos:].encode("utf-8")))
    ids = np.concatenate(parts)[: max_length - 1]
    if ids.size and ids[-1] == EOS_TOKEN_ID:  # "do not add eos again" (tokenization_byt5.py:136-145)
        return ids
    return np.concatenate([ids, np.array([EOS_TOKEN_ID], dtype=np.int32)])


def encode_packed(texts: Sequence[str], max_length: int) -> Tuple[np.ndarray, np.ndarray]:
    """Varlen form consumed by ``rp_encode_varlen``: (ids int32 [T], cu_seqlens int32 [B+1])."""
    rows = [encode_one(t, max_length) for t in texts]
    cu = np.zeros(len(rows) + 1, dtype=np.int32)
    if rows:
        cu[1:] = np.cumsum([r.size for r in rows])
        return np.concatenate(rows).astype(np.int32), cu
    return np.zeros(0, dtype=np.int32), cu


class BatchEncoding(dict):
    """Minimal stand-in for transformers' BatchEncoding: attribute access + ``.to(device)``."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def to(self, device):
        return BatchEncoding({k: v.to(device) if isinstance(v, torch.Tensor) else v for k, v in self.items()})


class ByT5Tokenizer:
    """Callable with the keyword surface the reference uses:
    ``tokenizer(texts, padding="longest", max_length=L, truncation=True, return_tensors="pt")``
    → ``.input_ids`` / ``.attention_mask`` int64 ``[B, longest]``."""

    pad_token_id = PAD_TOKEN_ID
    eos_token_id = EOS_TOKEN_ID
    unk_token_id = UNK_TOKEN_ID
    model_input_names = ["input_ids", "attention_mask"]

    def __call__(self, texts, padding="longest", max_length: int = 2048, truncation: bool = True,
                 return_tensors: str = "pt") -> BatchEncoding:
        if isinstance(texts, str):
            texts = [texts]
        assert padding == "longest" and truncation, "only the reference's call form is implemented"
        rows = [encode_one(t, max_length) for t in texts]
        L = max(r.size for r in rows)
        ids = np.zeros((len(rows), L), dtype=np.int64)
        mask = np.zeros((len(rows), L), dtype=np.int64)
        for i, r in enumerate(rows):
            ids[i, : r.size] = r
            mask[i, : r.size] = 1
        if return_tensors == "np":
            return BatchEncoding(input_ids=ids, attention_mask=mask)
        return BatchEncoding(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask))

    def packed(self, texts: Sequence[str], max_length: int) -> Tuple[np.ndarray, np.ndarray]:
        return encode_packed(texts, max_length)
