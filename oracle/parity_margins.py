"""TEST INFRASTRUCTURE (never imported by the product): the end-to-end parity MARGINS of the engine against the golden
fixtures the reference + HuggingFace produced in fp32 (tests/golden/g5, g7, g9; made by tests/golden/make_golden.py),
beside what HuggingFace's own bf16 mode - the reference's GPU numerics, retrieval/model.py:59-64 - reaches on the same
inputs.  One place computes them; the `-m gpu` tests assert on the numbers and write them to
profiles/r06_parity_margins.json (+ gpurun_out/), `__graft_entry__.smoke()` prints them.

The written contract: scores within 1e-2 absolute of the fp32 oracle, embeddings
cosine >= 0.999, ids equal wherever the oracle's gap to both neighbours exceeds 2 x tol, no worse than HF-bf16 on the same
inputs.  Two fixture families (reprover_amd/synth.py::synth_state_dict):
  * **G5h / G7h / G9h** - weights at exactly HF's init scales, the recipe SURVEY.md 8c gives for G5 and the one the
    contract's numbers were derived on.  The GPU tests assert the contract on them AS WRITTEN (`assert_written_contract`).
  * G5 / G7 / G9 - the deliberately sharp stress family (q 4x, position table 38x HF's scale), on which HF-bf16 itself
    misses 0.999 / 1e-2; there the tests enforce "no further from fp32 than HF-bf16 on any metric" + absolute floors, and
    `contract_met` records whether the written numbers happen to hold."""
import json
import os

import numpy as np
import torch


def gap_rule_ids(ours_ids, gold_ids, gold_scores, tol):
    """ids must agree at every rank whose golden score is separated from both neighbours by more
    than 2*tol.  Returns (#checked, #mismatched)."""
    checked = bad = 0
    for o, g, s in zip(ours_ids, gold_ids, gold_scores):
        s = np.asarray(s, dtype=np.float64)
        for r in range(len(g)):
            left = s[r - 1] - s[r] if r > 0 else np.inf
            right = s[r] - s[r + 1] if r + 1 < len(g) else 0.0  # the (k+1)-th is unknown: skip the last rank
            if left > 2 * tol and right > 2 * tol:
                checked += 1
                bad += int(o[r] != g[r])
    return checked, bad


def _cos(a, b):
    return torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=1)


def embedding_margins(emb: torch.Tensor, gold: torch.Tensor, hf_bf16: torch.Tensor) -> dict:
    """emb / gold / hf_bf16: [n, D] fp32 on the CPU (engine output, fp32 golden, HF-bf16 on the same inputs)."""
    cos, cos_hf = _cos(emb, gold), _cos(hf_bf16, gold)
    d = {
        "rows": int(emb.shape[0]),
        "min_row_cosine": float(cos.min()), "hf_bf16_min_row_cosine": float(cos_hf.min()),
        "max_abs_emb_err": float((emb - gold).abs().max()), "hf_bf16_max_abs_emb_err": float((hf_bf16 - gold).abs().max()),
        "max_abs_pairwise_score_err": float(((emb @ emb.T) - (gold @ gold.T)).abs().max()),
        "hf_bf16_max_abs_pairwise_score_err": float(((hf_bf16 @ hf_bf16.T) - (gold @ gold.T)).abs().max()),
        "rows_further_from_fp32_than_hf_bf16": int((cos < cos_hf - 1e-4).sum()),
        # the maximum over n (n - 1) / 2 pairs is a noisy statistic (it moves by +-10 % with any change of rounding order);
        # the root-mean-square over the same pairs is the robust form of "no further from fp32 than HF-bf16 on scores"
        "rms_pairwise_score_err": float(((emb @ emb.T) - (gold @ gold.T)).double().pow(2).mean().sqrt()),
        "hf_bf16_rms_pairwise_score_err": float(((hf_bf16 @ hf_bf16.T) - (gold @ gold.T)).double().pow(2).mean().sqrt()),
    }
    d["contract_met"] = bool(d["min_row_cosine"] >= 0.999 and d["max_abs_pairwise_score_err"] <= 1e-2)
    d["hf_bf16_meets_contract"] = bool(d["hf_bf16_min_row_cosine"] >= 0.999 and d["hf_bf16_max_abs_pairwise_score_err"] <= 1e-2)
    d["no_worse_than_hf_bf16"] = bool(d["rows_further_from_fp32_than_hf_bf16"] == 0
                                      and d["max_abs_emb_err"] <= d["hf_bf16_max_abs_emb_err"]
                                      and d["max_abs_pairwise_score_err"] <= d["hf_bf16_max_abs_pairwise_score_err"])
    return d


def assert_written_contract(m: dict) -> None:
    """The floating-point contract of BASELINE.md section 2 / SURVEY.md 8c exactly as written - no max(.., HF) escape, no
    lowered floor.  For embedding fixtures: every row's cosine >= 0.999, pairwise scores within 1e-2, and no metric worse
    than HF-bf16's on the same inputs.  For predict fixtures: max |d score| <= 1e-2, zero mismatches under the id gap rule,
    and the score error no larger than HF-bf16's."""
    if "min_row_cosine" in m and "max_abs_pairwise_score_err" in m:
        assert m["min_row_cosine"] >= 0.999, m
        assert m["max_abs_pairwise_score_err"] <= 1e-2, m
        assert m["rows_further_from_fp32_than_hf_bf16"] == 0, m
        assert m["max_abs_emb_err"] <= m["hf_bf16_max_abs_emb_err"], m
        assert m["max_abs_pairwise_score_err"] <= m["hf_bf16_max_abs_pairwise_score_err"], m
    elif "min_row_cosine" in m:
        assert m["min_row_cosine"] >= 0.999 and m["min_row_cosine"] >= m["hf_bf16_min_row_cosine"], m
    else:
        assert m["max_abs_score_err"] <= 1e-2 and m["gap_rule_mismatches"] == 0, m
        assert m["max_abs_score_err"] <= m["hf_bf16_max_abs_score_err"], m
    assert m["contract_met"], m


def g5_margins(model, golden_dir, fixture: str = "g5_byt5_small.npz") -> dict:
    """`model`: PremiseRetriever of the synthetic ByT5-small (fp32 outputs).  Fixture G5 (or "g5h_byt5_small.npz", the
    HF-init-scale family): 16 texts, 8 ... 2048 bytes."""
    g = np.load(os.path.join(golden_dir, fixture), allow_pickle=True)
    emb = model.encode_texts(list(g["texts"])).float().cpu()
    return embedding_margins(emb, torch.from_numpy(g["emb"]), torch.from_numpy(g["emb_hf_bf16"].astype(np.float32)))


def g9_margins(model, golden_dir, fixture: str = "g9_byt5_base.npz") -> dict:
    """`model`: PremiseRetriever of the synthetic ByT5-base, all 18 layers.  Fixture G9 (or "g9h_byt5_base.npz"): 8 texts."""
    g = np.load(os.path.join(golden_dir, fixture), allow_pickle=True)
    emb = model.encode_texts(list(g["texts"])).float().cpu()
    return embedding_margins(emb, torch.from_numpy(g["emb"]), torch.from_numpy(g["emb_hf_bf16"].astype(np.float32)))


def g7_corpus_records(g: dict):
    """The corpus.jsonl records of a G7-style fixture, regenerated from its seeds (G7: independent bodies; G7h: families)."""
    from reprover_amd import synth

    if g.get("corpus") == "family":
        return synth.synth_family_corpus_records(g["n_files"], g["n_premises"], seed=g["corpus_seed"],
                                                 code_bytes=tuple(g["code_bytes"]), family_size=g["family_size"])[0]
    return synth.synth_corpus_records(g["n_files"], g["n_premises"], seed=g["corpus_seed"], code_bytes=tuple(g["code_bytes"]))


def g7_row_margins(model, g, z) -> dict:
    """Every row of the re-indexed 1005-premise corpus (bf16 rows, the GPU default) against the reference's fp32 matrix."""
    Ef = model.corpus_embeddings.float().cpu()
    gold = torch.from_numpy(z["E_all_f16"].astype(np.float32))
    cos = _cos(Ef, gold)
    return {"rows": int(Ef.shape[0]), "min_row_cosine": float(cos.min()), "mean_row_cosine": float(cos.mean()),
            "hf_bf16_min_row_cosine": float(g["hf_bf16_min_embedding_cosine"]),
            "max_abs_emb_err": float((Ef - gold).abs().max()),
            "contract_met": bool(cos.min() >= 0.999)}


def g7_predict(model, g):
    """BASELINE configs[0] through the product API: the 128 states of G7 through predict_step in the fixture's batches.
    Returns (records, ids [128][k], scores ndarray, margins)."""
    from reprover_amd.common import Context, Pos

    k = g["k"]
    model.num_retrieved = k
    ctxs = [Context(q["path"], f"thm{j}", Pos(*q["pos"]), q["state"]) for j, q in enumerate(g["queries"])]
    where = {id(p): i for i, p in enumerate(model.corpus.all_premises)}
    model.predict_step_outputs = []
    for i in range(0, len(ctxs), g["batch_size"]):
        batch = ctxs[i : i + g["batch_size"]]
        tok = model.tokenizer([c.serialize() for c in batch], padding="longest", max_length=g["max_seq_len"],
                              truncation=True, return_tensors="pt")
        b = {"context": batch, "context_ids": tok.input_ids.cuda(), "context_mask": tok.attention_mask.cuda()}
        for key in ("url", "commit", "file_path", "full_name", "start", "tactic_idx", "all_pos_premises"):
            b[key] = [None] * len(batch)
        model.predict_step(b, 0)
    recs = model.predict_step_outputs
    ids = [[where[id(p)] for p in r["retrieved_premises"]] for r in recs]
    scores = np.array([r["scores"] for r in recs])
    gold_s = np.array(g["scores"])
    checked, bad = gap_rule_ids(ids, g["ids"], g["scores"], tol=1e-2)
    hf_err = float(np.abs(np.array(g["hf_bf16_scores_at_gold_ids"]) - gold_s).max())
    m = {
        "queries": len(ctxs), "k": k, "ranks": len(ctxs) * k,
        "max_abs_score_err": float(np.abs(scores - gold_s).max()), "hf_bf16_max_abs_score_err": hf_err,
        "gap_rule_ranks_checked": int(checked), "gap_rule_mismatches": int(bad),
        "top1_agreement": float(np.mean([a[0] == b[0] for a, b in zip(ids, g["ids"])])),
        f"top{k}_overlap": float(np.mean([len(set(a) & set(b)) / k for a, b in zip(ids, g["ids"])])),
    }
    if "hf_bf16_predict_ids" in g:  # what the reference's own GPU mode (HF-bf16 + bf16 similarity matrix) retrieves
        hf_ids = g["hf_bf16_predict_ids"]
        _, hf_bad = gap_rule_ids(hf_ids, g["ids"], g["scores"], tol=1e-2)
        m["hf_bf16_top1_agreement"] = float(np.mean([a[0] == b[0] for a, b in zip(hf_ids, g["ids"])]))
        m[f"hf_bf16_top{k}_overlap"] = float(np.mean([len(set(a) & set(b)) / k for a, b in zip(hf_ids, g["ids"])]))
        m["hf_bf16_gap_rule_mismatches"] = int(hf_bad)
    m["contract_met"] = bool(m["max_abs_score_err"] <= 1e-2 and bad == 0)
    m["hf_bf16_meets_contract"] = bool(hf_err <= 1e-2)
    return recs, ids, scores, ctxs, m


def write_margins(margins: dict, root: str) -> None:
    """profiles/r06_parity_margins.json (tracked) and gpurun_out/parity_margins.json (what travels back from the GPU box)."""
    doc = {
        "what": "end-to-end parity margins of the HIP engine vs the fp32 goldens of the reference + HuggingFace "
                "(tests/golden), beside HF-bf16's own error on the same inputs; written by `pytest -m gpu`",
        "written_contract": "scores within 1e-2 abs, embedding cosine >= 0.999, ids equal where the oracle's rank gap > 2e-2 "
                            "(BASELINE.md section 2)",
        "enforced_bar": "fixtures g5h / g7h / g9h (weights at HF's init scales, SURVEY.md 8c's recipe): the written contract "
                        "as it stands + no metric worse than HF-bf16's.  Fixtures g5 / g7 / g9 (sharp stress family, on which "
                        "HF-bf16 itself misses 0.999 / 1e-2): no row and no metric further from fp32 than HF-bf16, max |d score| "
                        "<= max(1e-2, HF-bf16's), cosine floors 0.997 (12 layers) / 0.99 (18 layers), gap-rule mismatches = 0",
        **margins,
    }
    for path in (os.path.join(root, "profiles", "r06_parity_margins.json"), os.path.join(root, "gpurun_out", "parity_margins.json")):
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as fh:
                json.dump(doc, fh, indent=1, sort_keys=True)
                fh.write("\n")
        except OSError:
            pass
