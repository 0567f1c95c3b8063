"""Deterministic synthetic inputs: encoder weights, premise/state texts, corpus files.

There is no network in the build or GPU environments, so neither the pretrained
``kaiyuy/leandojo-lean4-retriever-byt5-small`` checkpoint nor the LeanDojo benchmark can be
fetched.  Tests, ``bench.py`` and the golden-vector generator all draw from the generators here
(seed 3407 = the reference's ``seed_everything``, retrieval/confs/cli_lean4_random.yaml:1), which
are bit-reproducible from raw Philox counters so that the authoring container and the GPU box
build identical weights without shipping a 0.9 GB checkpoint.
"""
from __future__ import annotations

import json
import zlib
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

SEED = 3407

CONFIGS: Dict[str, Dict] = {
    # google/byt5-small encoder
    "byt5-small": dict(vocab_size=384, d_model=1472, d_kv=64, num_heads=6, d_ff=3584, num_layers=12),
    # google/byt5-base encoder
    "byt5-base": dict(vocab_size=384, d_model=1536, d_kv=64, num_heads=12, d_ff=3968, num_layers=18),
    # small shapes for fast parity tests; keeps d_kv = 64 like both real models
    "tiny": dict(vocab_size=384, d_model=128, d_kv=64, num_heads=2, d_ff=256, num_layers=2),
}
for _c in CONFIGS.values():
    _c.update(
        relative_attention_num_buckets=32,
        relative_attention_max_distance=128,
        layer_norm_epsilon=1e-6,
        feed_forward_proj="gated-gelu",
    )


def t5_config(name: str) -> Dict:
    return dict(CONFIGS[name])


def _philox_normal(name: str, n: int, seed: int) -> np.ndarray:
    """n standard normals from raw Philox-4x64 output (stable by specification) via Box-Muller
    in float64; the stream is keyed by (seed, crc32(name)) so tensors are independent."""
    key = (int(seed) << 32) ^ zlib.crc32(name.encode())
    raw = np.random.Philox(key=key).random_raw(2 * ((n + 1) // 2))
    u = (raw >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    u1, u2 = u[0::2], u[1::2]
    r = np.sqrt(-2.0 * np.log1p(-u1))  # u1 in [0,1) -> log argument in (0,1]
    z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])
    return z[:n]


def _philox_uniform(name: str, n: int, seed: int) -> np.ndarray:
    key = (int(seed) << 32) ^ zlib.crc32(name.encode())
    raw = np.random.Philox(key=key).random_raw(n)
    return (raw >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


SCALES = ("sharp", "hf")


def synth_state_dict(cfg: Dict, seed: int = SEED, scale: str = "sharp") -> Dict[str, torch.Tensor]:
    """HF-keyed fp32 encoder state dict (key names: SURVEY.md App. B.5).

    Two families, the same Philox normals under two sets of standard deviations:

    * ``scale="hf"`` - exactly HF's T5 init scales (modeling_t5.py:563-616, SURVEY.md section 8c's G5 recipe):
      q ~ N(0, (D·dk)^-1), k / v / wi ~ N(0, 1/D), o ~ N(0, 1/(H·dk)), wo ~ N(0, 1/F), embedding ~ N(0, 1),
      relative-position table ~ N(0, 1/D); layer-norm weights U(0.5, 1.5) as the survey's recipe says.  These are the
      weights the WRITTEN floating-point contract (scores within 1e-2, cosine >= 0.999) was derived on; the "h"
      fixtures (G5h / G7h / G9h) use them and the GPU tests assert the contract on them as written.
    * ``scale="sharp"`` (default; fixtures G5 / G7 / G9) - the stress family: q 4x and the relative-position table
      ~38x HF's scale (q ~ N(0, (0.5·D^-½)²) → logits std ≈ 4; table ~ N(0, 1)), so that attention is sharp and
      mask / bias / scale bugs cannot hide behind a near-uniform softmax.  HF-bf16 itself misses the written
      contract on this family.
    """
    if scale not in SCALES:
        raise ValueError(f"scale must be one of {SCALES}, got {scale!r}")
    D, dk, H, F, V = cfg["d_model"], cfg["d_kv"], cfg["num_heads"], cfg["d_ff"], cfg["vocab_size"]
    inner = H * dk
    q_std = (D * dk) ** -0.5 if scale == "hf" else 0.5 * D**-0.5
    table_std = D**-0.5 if scale == "hf" else 1.0
    sd: Dict[str, torch.Tensor] = {}

    def normal(name, shape, std):
        n = int(np.prod(shape))
        sd[name] = torch.from_numpy((_philox_normal(name, n, seed) * std).astype(np.float32).reshape(shape))

    def ln(name):
        sd[name] = torch.from_numpy((0.5 + _philox_uniform(name, D, seed)).astype(np.float32))

    normal("shared.weight", (V, D), 1.0)
    sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
    normal(
        "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight",
        (cfg["relative_attention_num_buckets"], H),
        table_std,
    )
    for i in range(cfg["num_layers"]):
        p = f"encoder.block.{i}.layer."
        ln(p + "0.layer_norm.weight")
        normal(p + "0.SelfAttention.q.weight", (inner, D), q_std)
        normal(p + "0.SelfAttention.k.weight", (inner, D), D**-0.5)
        normal(p + "0.SelfAttention.v.weight", (inner, D), D**-0.5)
        normal(p + "0.SelfAttention.o.weight", (D, inner), inner**-0.5)
        ln(p + "1.layer_norm.weight")
        normal(p + "1.DenseReluDense.wi_0.weight", (F, D), D**-0.5)
        normal(p + "1.DenseReluDense.wi_1.weight", (F, D), D**-0.5)
        normal(p + "1.DenseReluDense.wo.weight", (D, F), F**-0.5)
    ln("encoder.final_layer_norm.weight")
    return sd


# ----------------------------------------------------------------------------------------------
# Texts
# ----------------------------------------------------------------------------------------------
_SYMS = ["ℕ", "⊢", "→", "∀"]  # 3-byte UTF-8 symbols (SURVEY.md §8d byte-value mix)


def synth_text(rng: np.random.Generator, n_bytes: int) -> str:
    """A string of exactly ``n_bytes`` UTF-8 bytes: ≈90 % printable ASCII, ≈10 % of the bytes
    belong to 3-byte math symbols.  No '<' so no tokenizer special can appear by accident."""
    out: List[str] = []
    left = n_bytes
    while left > 0:
        if left >= 3 and rng.random() < 0.035:
            out.append(_SYMS[int(rng.integers(4))])
            left -= 3
        else:
            c = int(rng.integers(32, 127))
            if c == 60:  # '<'
                c = 32
            out.append(chr(c))
            left -= 1
    return "".join(out)


def synth_lengths(rng: np.random.Generator, n: int, kind: str = "mix", lo: int = 8, hi: int = 2048) -> np.ndarray:
    """Token lengths incl. EOS.  'mix' = clip(round(LogNormal(ln 180, 0.9)), lo, hi) (the
    mathlib-like mix of SURVEY.md §8d); an integer string = fixed tier."""
    if kind == "mix":
        return np.clip(np.rint(rng.lognormal(np.log(180.0), 0.9, size=n)), lo, hi).astype(np.int64)
    return np.full(n, int(kind), dtype=np.int64)


def synth_token_batch(
    rng: np.random.Generator, lengths: Sequence[int]
) -> Tuple[np.ndarray, np.ndarray]:
    """Packed ByT5 ids for sequences of the given lengths (each ends in EOS = 1) without going
    through strings: ids uniform over printable-ASCII byte ids.  Returns (ids int32 [T],
    cu_seqlens int32 [B+1])."""
    lengths = np.asarray(lengths, dtype=np.int64)
    cu = np.zeros(len(lengths) + 1, dtype=np.int32)
    cu[1:] = np.cumsum(lengths)
    ids = rng.integers(32 + 3, 127 + 3, size=int(cu[-1]), dtype=np.int32)
    ids[cu[1:] - 1] = 1
    return ids, cu


# ----------------------------------------------------------------------------------------------
# Corpus (corpus.jsonl of SURVEY.md App. B.1)
# ----------------------------------------------------------------------------------------------
def synth_corpus_records(
    n_files: int,
    n_premises: int,
    seed: int = SEED,
    code_bytes: Tuple[int, int] = (24, 120),
    max_imports: int = 3,
    with_edge_cases: bool = True,
    body_fn=None,
) -> List[dict]:
    """File records in topological order.  Premises get dotted names, code that mentions the
    name (so ``serialize`` has something to mark), increasing positions inside a file with some
    nested (non-monotone ``end``) declarations, and — when ``with_edge_cases`` — the records
    ``File.from_data`` must drop (null / ``user__.n`` / ``[mutual]`` names, empty code) plus
    duplicate ``full_name``s inside a file."""
    rng = np.random.default_rng(seed)
    weights = rng.lognormal(0.0, 0.8, size=n_files)
    counts = np.floor(weights / weights.sum() * n_premises).astype(int)
    counts[: n_premises - counts.sum()] += 1
    files = []
    for f in range(n_files):
        path = f"Synth/Dir{f % 7}/File{f}.lean"
        n_imp = int(rng.integers(0, max_imports + 1)) if f > 0 else 0
        imports = sorted({int(x) for x in rng.integers(max(0, f - 40), f, size=n_imp)}) if f > 0 else []
        prem = []
        line = 1
        for j in range(int(counts[f])):
            ns = ["Nat", "List", "Set", "Finset", "Real"][int(rng.integers(5))]
            short = f"lemma_{f}_{j}"
            full = f"{ns}.{short}" if rng.random() < 0.8 else short
            nbytes = int(rng.integers(code_bytes[0], code_bytes[1]))
            style = rng.random()
            if style < 0.5:
                head = f"theorem {short} "
            elif style < 0.7:
                head = f"theorem _root_.{full} "
            elif style < 0.85:
                head = f"lemma «{short}» "
            else:
                head = "instance : Inhabited Foo "
            body = synth_text(rng, max(4, nbytes - len(head.encode())))
            if body_fn is not None:  # the main stream above is consumed either way: structure and positions do not move
                body = body_fn(f, j, body)
            start = [line, int(rng.integers(0, 4))]
            span = int(rng.integers(1, 6))
            end = [line + span, int(rng.integers(0, 40))]
            # sometimes nest the next declaration inside this one (end not monotone in order)
            line += span + 1 if rng.random() < 0.85 else 1
            prem.append({"full_name": full, "code": head + ": " + body, "start": start, "end": end, "kind": "theorem"})
        if with_edge_cases and f % 9 == 3:
            prem.insert(0, {"full_name": None, "code": "x", "start": [1, 0], "end": [1, 1], "kind": "x"})
            prem.append({"full_name": "user__.n.1", "code": "y", "start": [line, 0], "end": [line, 1], "kind": "x"})
            prem.append({"full_name": "[mutual.a, mutual.b]", "code": "z", "start": [line, 0], "end": [line, 1], "kind": "x"})
            prem.append({"full_name": f"Empty.code_{f}", "code": "", "start": [line, 0], "end": [line, 1], "kind": "x"})
        if with_edge_cases and f % 11 == 5 and len(prem) >= 2 and prem[0]["full_name"] is not None:
            dup = dict(prem[0])
            dup["start"], dup["end"] = [line + 1, 0], [line + 2, 0]
            dup["code"] = dup["code"] + " -- again"
            prem.append(dup)
        files.append({"path": path, "imports": [files[i]["path"] for i in imports], "premises": prem})
    return files


def corrupt_text(rng: np.random.Generator, s: str, rate: float) -> str:
    """``s`` with round(rate * len) of its characters replaced by random printable ASCII (never '<')."""
    chars = list(s)
    for i in rng.choice(len(chars), size=int(round(rate * len(chars))), replace=False):
        c = int(rng.integers(33, 127))
        chars[i] = " " if c == 60 else chr(c)
    return "".join(chars)


def synth_family_corpus_records(n_files: int, n_premises: int, seed: int, code_bytes: Tuple[int, int] = (24, 120),
                                family_size: int = 12, **kw) -> Tuple[List[dict], List[dict]]:
    """``synth_corpus_records`` (same files, imports, names, positions, edge cases) whose premise bodies come in FAMILIES
    of near-duplicates: ``family_size`` consecutive premises of a file share one base text, member i carrying it with
    a fraction ``rate_i`` of its characters replaced (the ladder 0, 1/n, 2/n, ... in a seeded order) - the way a library
    holds ``foo``, ``foo'``, ``foo_left`` ....  A state built from a family's base text then meets retrieval scores
    spread from ~0.95 down to the corpus background in steps of several 1e-2, so that the id comparison of the
    end-to-end parity fixture G7h (ids must agree wherever the oracle's rank gap exceeds 2 x tol) has ranks to check;
    independent random bodies give gaps of a few 1e-3 and almost none.

    Returns (records, families); families[g] = {"file": f, "base": str, "members": [full_name, ...], "rates": [...]}."""
    families: Dict[Tuple[int, int], dict] = {}

    def body_fn(f, j, body):
        key = (f, j // family_size)
        fam = families.get(key)
        if fam is None:
            frng = np.random.default_rng([seed, 77, f, j // family_size])
            base = synth_text(frng, int(frng.integers(max(code_bytes[0], 40), max(code_bytes[1], 41))))
            fam = families[key] = {"file": f, "base": base, "order": frng.permutation(family_size), "slots": [], "rng": frng}
        rate = float(fam["order"][j % family_size]) / family_size
        fam["slots"].append((j, rate))
        return corrupt_text(fam["rng"], fam["base"], rate)

    records = synth_corpus_records(n_files, n_premises, seed=seed, code_bytes=code_bytes, body_fn=body_fn, **kw)
    out = []
    for (f, _), fam in sorted(families.items()):
        # names of the members: the j-th premise GENERATED for file f (edge-case records are inserted around them)
        gen = [p for p in records[f]["premises"] if p["full_name"] and p["full_name"].split(".")[-1].startswith(f"lemma_{f}_")
               and not p["code"].endswith(" -- again")]
        by_j = {int(p["full_name"].rsplit("_", 1)[1]): p["full_name"] for p in gen}
        out.append({"file": f, "base": fam["base"], "members": [by_j[j] for j, _ in fam["slots"]],
                    "rates": [r for _, r in fam["slots"]]})
    return records, out


def write_corpus_jsonl(path: str, records: Sequence[dict]) -> None:
    with open(path, "w") as fh:
        for rec in records:
            fh.write(json.dumps(rec, ensure_ascii=False) + "\n")


def synth_state(rng: np.random.Generator, n_bytes: int) -> str:
    """A proof-state-like string containing the mandatory '⊢' (common.py:46-52)."""
    n_bytes = max(n_bytes, 8)
    head = synth_text(rng, (n_bytes - 4) // 2)
    tail = synth_text(rng, n_bytes - 4 - len(head.encode()))
    return head + " ⊢" + tail  # ' ' + 3-byte turnstile = 4 bytes


# ----------------------------------------------------------------------------------------------
# Accessibility operands without a corpus (sim-only benches and kernel tests at sizes no corpus.jsonl is built for)
# ----------------------------------------------------------------------------------------------
def synth_masks(rng, N, B, F, density=0.3):
    """Random accessibility operands + the boolean [B, N] predicate they encode."""
    cuts = np.sort(rng.choice(np.arange(1, N), size=F - 1, replace=False)) if F > 1 else np.array([], dtype=int)
    file_of = np.zeros(N, dtype=np.int32)
    file_of[cuts] = 1
    file_of = np.cumsum(file_of).astype(np.int32)
    end_key = rng.integers(0, 1 << 30, size=N).astype(np.int64)
    own = rng.integers(0, F, size=B).astype(np.int32)
    qk = rng.integers(0, 1 << 30, size=B).astype(np.int64)
    imp = rng.random((B, F)) < density
    imp[np.arange(B), own] = False
    words = (B + 31) // 32
    padded = np.zeros((F, words * 32), dtype=np.uint8)
    padded[:, :B] = imp.T
    bits_t = np.packbits(padded, axis=1, bitorder="little").view(np.uint32).reshape(F, words)
    acc = imp[:, file_of] | ((file_of[None, :] == own[:, None]) & (end_key[None, :] <= qk[:, None]))
    return (file_of, end_key, bits_t, own, qk), acc



# ----------------------------------------------------------------------------------------------
# Dataset splits ({train,val,test}.json of SURVEY.md App. B.2)
# ----------------------------------------------------------------------------------------------
def synth_split(records: Sequence[dict], n_theorems: int, seed: int, min_file: int = 0, accept=None) -> List[dict]:
    """Theorems with traced tactics whose annotated tactics point at premises of the corpus records
    (positions inside real premises, plus some that resolve to nothing and some tactics with no
    premises at all)."""
    rng = np.random.default_rng(seed)
    usable = [i for i, r in enumerate(records) if i >= min_file]
    out = []
    for t in range(n_theorems):
        f = int(rng.choice(usable))
        start = [int(rng.integers(1, 400)), int(rng.integers(0, 30))]
        while accept is not None and not accept(records[f]["path"], start):  # e.g. "enough accessible premises"
            f = int(rng.choice(usable))
            start = [int(rng.integers(1, 400)), int(rng.integers(0, 30))]
        rec = records[f]
        tactics = []
        for k in range(int(rng.integers(1, 5))):
            provs = []
            for _ in range(int(rng.integers(0, 4))):
                g = int(rng.integers(0, f + 1))
                prem = [p for p in records[g]["premises"] if p["full_name"]]
                if prem and rng.random() < 0.85:
                    p = prem[int(rng.integers(len(prem)))]
                    pos = [int(p["start"][0]), int(p["start"][1])]
                else:
                    pos = [100000, 0]  # resolves to no premise
                provs.append({"full_name": "x", "def_path": records[g]["path"], "def_pos": pos, "def_end_pos": pos})
            state = synth_state(rng, int(rng.integers(30, 160)))
            tactics.append({"tactic": f"simp [l{k}]", "annotated_tactic": [f"simp [<a>l{k}</a>]", provs],
                            "state_before": state, "state_after": "no goals"})
        out.append({"url": "https://example.org/synth", "commit": "0" * 40, "file_path": rec["path"],
                    "full_name": f"Synth.thm_{seed}_{t}", "start": start, "end": [start[0] + 3, 0],
                    "traced_tactics": tactics})
    return out
