"""Benchmark of the retrieval hot path on MI355X (driver contract: see the task statement).

One *step* = one pass of the hot path over one batch of synthetic input, BASELINE.json configs[1]:
encode B = 256 proof states with the ByT5-small encoder (random-init weights of that
architecture) and retrieve the top-100 accessible premises for each from a resident
130,000-premise bf16 index (similarity GEMM + accessibility mask + exact top-k).  Inputs (token
ids, masks, the index) are resident in HBM when the timed region starts.

N > 1 GPUs (one process per GPU, torch.distributed over RCCL): the index is row-sharded N ways;
every rank encodes its own 256 states (weak scaling), query embeddings are all-gathered, each
rank scans its shard for all N*256 queries, the per-shard top-k lists travel as ONE packed all-gather
([scores | ids | counts] per rank) and each rank merges the lists of its own queries straight from the receive buffer
(two collectives per step in all: query embeddings, result block).  value = (N*256 queries) / max-over-ranks step time.

The timed region is un-instrumented; a second pass of the same K steps with an event pair around every launch gives
the per-kernel split the roofline objects are computed from.  Beside the headline value the line carries, at N = 1
(SURVEY.md §8d): the same queries through the PRODUCT API from strings (`product_api_qps`: 16 eval batches of 64 states,
RetrievalDataset-style collate -> PremiseRetriever.predict_step, which gathers host batches into 256-state GPU passes;
`product_api` also holds the 4-batch figure and the figure with one pass per batch), premises/s at the fixed length tiers 128 / 512 /
2048 and on the length mix, the wall time of a full 130,000-premise `reindex_corpus`, the wall latency of
single-state `retrieve()` calls, and the CPU baseline (the oracle on the host cores, bounded sample).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from reprover_amd import _lib, build, synth  # noqa: E402
from reprover_amd.common import Context, Corpus, Pos  # noqa: E402
from reprover_amd.encoder import HipT5Encoder  # noqa: E402

N_PREMISES, N_FILES, B_STATES, TOP_K = 130_000, 5_000, 256, 100
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0  # HBM3E spec


def fast_corpus_records(n_files: int, n_premises: int, seed: int):
    """corpus.jsonl records with a dense-ish import DAG (transitive closure of a typical late file
    covers a large fraction of earlier files, like mathlib) and trivial code strings."""
    rng = np.random.default_rng(seed)
    w = rng.lognormal(0.0, 0.8, size=n_files)
    counts = np.floor(w / w.sum() * n_premises).astype(int)
    counts[: n_premises - counts.sum()] += 1
    files = []
    for f in range(n_files):
        n_imp = int(rng.integers(1, 5)) if f else 0
        imps = sorted({int(x) for x in rng.integers(max(0, f - 300), f, size=n_imp)}) if f else []
        prem, line = [], 1
        for j in range(int(counts[f])):
            prem.append({"full_name": f"F{f}.t{j}", "code": f"theorem t{j} : True", "start": [line, 0],
                         "end": [line + 1, 10]})
            line += 2
        files.append({"path": f"M/F{f}.lean", "imports": [files[i]["path"] for i in imps], "premises": prem})
    return files


def random_init_state_dict(cfg, device, seed):
    """Random-init ByT5-small encoder weights (HF init scales), generated on the device."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    D, dk, H, F, V = cfg["d_model"], cfg["d_kv"], cfg["num_heads"], cfg["d_ff"], cfg["vocab_size"]
    inner = H * dk

    def n(shape, std):
        return torch.randn(shape, generator=g, device=device) * std

    sd = {"shared.weight": n((V, D), 1.0),
          "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight": n((32, H), 1.0),
          "encoder.final_layer_norm.weight": 0.5 + torch.rand(D, generator=g, device=device)}
    for i in range(cfg["num_layers"]):
        p = f"encoder.block.{i}.layer."
        sd[p + "0.layer_norm.weight"] = 0.5 + torch.rand(D, generator=g, device=device)
        sd[p + "0.SelfAttention.q.weight"] = n((inner, D), 0.5 * D ** -0.5)
        sd[p + "0.SelfAttention.k.weight"] = n((inner, D), D ** -0.5)
        sd[p + "0.SelfAttention.v.weight"] = n((inner, D), D ** -0.5)
        sd[p + "0.SelfAttention.o.weight"] = n((D, inner), inner ** -0.5)
        sd[p + "1.layer_norm.weight"] = 0.5 + torch.rand(D, generator=g, device=device)
        sd[p + "1.DenseReluDense.wi_0.weight"] = n((F, D), D ** -0.5)
        sd[p + "1.DenseReluDense.wi_1.weight"] = n((F, D), D ** -0.5)
        sd[p + "1.DenseReluDense.wo.weight"] = n((D, F), F ** -0.5)
    return sd


def fast_text_corpus_records(n_files: int, n_premises: int, seed: int):
    """The same DAG as fast_corpus_records, with premise code of the mathlib-like length mix (SURVEY.md §8d:
    clip(round(LogNormal(ln 180, 0.9)), 8, 2048) tokens incl. the <a></a> mark-up and EOS) - the input of the
    full re-index leg.  Vectorised: 130k strings from one random byte buffer."""
    recs = fast_corpus_records(n_files, n_premises, seed)
    rng = np.random.default_rng(seed + 5)
    n = sum(len(r["premises"]) for r in recs)
    lens = np.maximum(synth.synth_lengths(rng, n, "mix", lo=40, hi=2048) - 32, 8)  # room for name + mark-up + EOS
    buf = rng.integers(32, 127, size=int(lens.sum()), dtype=np.uint8)
    buf[buf == 60] = 32  # no '<': no tokenizer special by accident
    cu = np.concatenate([[0], np.cumsum(lens)])
    i = 0
    for r in recs:
        for pr in r["premises"]:
            body = buf[cu[i] : cu[i + 1]].tobytes().decode("ascii")
            pr["code"] = "theorem " + pr["full_name"].split(".")[-1] + " : " + body
            i += 1
    return recs


def _cpu_budget():
    """What this process may actually use of the host: the scheduler affinity, the cgroup CPU quota (v2 `cpu.max`, v1
    `cpu.cfs_quota_us`) and the physical cores behind the affinity mask (SMT siblings counted once).  torch's default thread
    count is the number of logical CPUs of the MACHINE; under a quota or a narrower affinity that oversubscribes the
    cores and the oracle runs many times slower than the host can (472 tok/s on "128 cores")."""
    try:
        aff = sorted(os.sched_getaffinity(0))
    except AttributeError:
        aff = list(range(os.cpu_count() or 1))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    phys = set()
    try:
        for c in aff:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            phys.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
    except OSError:
        phys = set()
    usable = len(aff)
    if quota is not None:
        usable = max(1, min(usable, int(quota + 0.5)))
    return {"cpus_online": os.cpu_count(), "cores_visible": len(aff), "physical_cores_visible": len(phys) or None,
            "cpu_quota": quota, "threads_usable": usable}


def cpu_baseline(cfg, sd_dev, corpus_path, E_dev, state_texts, state_ctx, n_encode=16, b_retrieve=256, n_premises=512,
                 premise_budget_s=130.0):
    """The reference's CPU path as restated by the oracle (kind "port"), timed on this host: pad-to-longest batch
    encode in fp32 at the reference's own matmul precision ("medium", retrieval/model.py:26; median of 3 passes) and at
    "highest" (one pass), and `get_nearest_premises` (Q @ E.T, full argsort, per-query Python accessibility walk:
    common.py:299-326) over the full 130k x 1472 fp32 matrix at the step's B = 256 (median of 5) and at B = 1 (median
    of 5).  A bounded sample of the step's workload (the 256 states of a step would take minutes).

    Threads: the count is taken from what the process may use (`_cpu_budget`: affinity, cgroup quota), and a short sweep
    (one encode pass of 4 states at each of up to four thread counts) picks the best one before the timed passes; the
    line reports the budget, the sweep and the count used (`cores`)."""
    from oracle import common_ref, t5_ref

    budget = _cpu_budget()
    default_threads = torch.get_num_threads()
    usable = budget["threads_usable"]
    phys = budget["physical_cores_visible"] or usable
    cands = sorted({t for t in (8, 32, min(phys, usable), usable) if 1 <= t <= usable})
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:  # numpy's BLAS then keeps its own default
        threadpool_limits = None

    sd = {k: v.float().cpu() for k, v in sd_dev.items()}
    E = E_dev.float().cpu().numpy()
    ref_corpus = common_ref.CorpusRef(corpus_path)
    texts = state_texts[:n_encode]
    ctxs = [common_ref.ContextRef(c.path, c.theorem_full_name, common_ref.Pos(*c.theorem_pos), c.state)
            for c in state_ctx[:b_retrieve]]
    padded = lambda tx: len(tx) * min(2048, max(len(s.encode()) + 1 for s in tx))
    torch.set_float32_matmul_precision("medium")
    sweep = {}
    for t in cands:  # one pass of 4 states per candidate (the first one also warms the allocator up: run it twice)
        torch.set_num_threads(t)
        for _ in range(2 if not sweep else 1):
            t0 = time.perf_counter()
            t5_ref.encode_texts(cfg, sd, texts[:4], 2048)
            sweep[t] = padded(texts[:4]) / (time.perf_counter() - t0)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    enc_s = {}
    q = None
    for prec, reps in (("medium", 3), ("highest", 1)):
        torch.set_float32_matmul_precision(prec)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            q = t5_ref.encode_texts(cfg, sd, texts, 2048).numpy()
            ts.append(time.perf_counter() - t0)
        enc_s[prec] = float(np.median(ts))
    # The metric's FIRST half - premises encoded/s - on the CPU as SURVEY.md 8(d) specifies it: the 128-byte length tier,
    # >= 512 premises, pad-to-longest batches of 64 (retrieval/index.py:24), the reference's "medium" precision, median of
    # up to 3 sweeps - bounded by a wall-clock budget so that the default run still ends within minutes (a sweep is ~40 s
    # on 16 threads; the number of sweeps actually timed is reported).
    torch.set_float32_matmul_precision("medium")
    rngp = np.random.default_rng(synth.SEED + 128)
    prem_texts = [synth.synth_text(rngp, 127) for _ in range(n_premises)]
    prem_ts, t_budget = [], time.perf_counter()
    while len(prem_ts) < 3 and (not prem_ts or time.perf_counter() - t_budget + prem_ts[-1] < premise_budget_s):
        t0 = time.perf_counter()
        t5_ref.encode_texts(cfg, sd, prem_texts, 2048, 64)
        prem_ts.append(time.perf_counter() - t0)
    prem_s = float(np.median(prem_ts))
    torch.set_float32_matmul_precision("highest")
    n_tok_padded = padded(texts)
    rngq = np.random.default_rng(11)
    Q = rngq.standard_normal((b_retrieve, E.shape[1])).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    Q[: len(q)] = q

    def timed(fn, reps):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    import contextlib

    with (threadpool_limits(limits=usable) if threadpool_limits else contextlib.nullcontext()):
        ret_b = timed(lambda: ref_corpus.get_nearest_premises(E, ctxs, Q, TOP_K), 5)
        ret_1 = timed(lambda: ref_corpus.get_nearest_premises(E, ctxs[:1], Q[:1], TOP_K), 5)
    torch.set_num_threads(default_threads)
    enc_q = n_encode / enc_s["medium"]
    ret_q = b_retrieve / ret_b
    return {
        "value": 1.0 / (1.0 / enc_q + 1.0 / ret_q),
        "unit": "queries/s",
        "cores": best,
        "threads_used": best,
        "torch_default_threads": default_threads,
        **budget,
        "encode_thread_sweep_tok_per_s": {str(k): v for k, v in sweep.items()},
        "encode_tok_per_s": {"medium": n_tok_padded / enc_s["medium"], "highest": n_tok_padded / enc_s["highest"]},
        "kind": "port",
        "sample": f"encode: {n_encode} of the step's 256 states as one pad-to-longest batch ({n_tok_padded} padded tokens) on "
                  f"{best} threads (best of a sweep over {cands}), fp32: medium median of 3 ({enc_s['medium']:.1f}s), highest "
                  f"once ({enc_s['highest']:.1f}s); retrieve: get_nearest_premises on the full 130k x 1472 fp32 index, "
                  f"B={b_retrieve} median of 5 ({ret_b:.2f}s), B=1 median of 5 ({ret_1 * 1e3:.0f}ms), BLAS limited to "
                  f"{usable} threads; value = encode(medium) and retrieve(B={b_retrieve}) rates combined per query",
        "deviation_from_survey_8d": "SURVEY.md 8(d) asks >= 512 premises per length tier x 3 repeats for the CPU encode: done "
                                    "for the 128-byte tier (premises_per_s, up to 3 sweeps inside a 130-s budget); the 512- and "
                                    "2048-byte tiers would add tens of minutes and follow from tokens/s (the CPU path is "
                                    "compute-bound: tokens/s is flat in the length up to the L^2 attention term); the state-encode "
                                    "leg times 16 states x 3 repeats (retrieve: 5 repeats as specified)",
        "premises_per_s": n_premises / prem_s,
        "premises_per_s_sample": f"{n_premises} premises of the 128-byte tier (127 bytes + EOS), pad-to-longest batches of 64, "
                                 f"fp32 'medium', {best} threads: median of {len(prem_ts)} sweep(s) of {prem_s:.1f}s "
                                 f"({n_premises * 128 / prem_s:.0f} tokens/s) - SURVEY.md 8(d)'s CPU encode figure for the 128-byte tier",
        "encode_qps_medium": enc_q,
        "encode_qps_highest": n_encode / enc_s["highest"],
        "retrieve_only_qps": ret_q,
        "retrieve_b1_latency_ms": ret_1 * 1e3,
        "cpu_model": _cpu_model(),
    }


def shard_shape_call_us(lib, corpus, E_full, dev):
    """Per-rank `rp_sim_topk` call at the N-GPU step's shard shapes, on ONE GPU: N x 256 queries (what the query all-gather
    hands every rank) against a 130,000 / N-row shard, N = 2, 4, 8 - so the scan stage's scaling can be read off while no
    multi-GPU node is available: ideal weak scaling keeps the call at the N = 1 time.  Random unit queries; the shard's own
    accessibility arrays."""
    out = {}
    N_all, D = E_full.shape
    for n in (1, 2, 4, 8):
        BQ, rows = B_STATES * n, N_all // n
        g = torch.Generator(device=dev)
        g.manual_seed(synth.SEED + 50 + n)
        Q = torch.nn.functional.normalize(torch.randn(BQ, D, generator=g, device=dev), dim=1).to(torch.bfloat16)
        rng = np.random.default_rng(synth.SEED + 60 + n)
        ctx = [Context(f"M/F{int(rng.integers(N_FILES // 2, N_FILES))}.lean", f"s{j}", Pos(int(rng.integers(1, 60)), 0), "⊢ True")
               for j in range(BQ)]
        bits_t, own, qk = corpus.query_masks(ctx)
        bits_d = torch.from_numpy(bits_t.view(np.int32)).to(dev)
        own_d, qk_d = torch.from_numpy(own).to(dev), torch.from_numpy(qk).to(dev)
        fo = torch.from_numpy(corpus.file_of[:rows].copy()).to(dev)
        ek = torch.from_numpy(corpus.end_key[:rows].copy()).to(dev)
        E = E_full[:rows]
        o_s = torch.empty((BQ, TOP_K), dtype=torch.float32, device=dev)
        o_i = torch.empty((BQ, TOP_K), dtype=torch.int32, device=dev)
        o_c = torch.empty((BQ,), dtype=torch.int32, device=dev)
        wb = lib.rp_sim_topk_workspace_bytes(BQ, rows, D, TOP_K, 0)
        ws = torch.empty(wb, dtype=torch.uint8, device=dev)

        def call():
            _lib.check(lib.rp_sim_topk(Q.data_ptr(), E.data_ptr(), BQ, rows, D, fo.data_ptr(), ek.data_ptr(), bits_d.data_ptr(),
                                       corpus.num_files, own_d.data_ptr(), qk_d.data_ptr(), 0, TOP_K, 0, o_s.data_ptr(),
                                       o_i.data_ptr(), o_c.data_ptr(), ws.data_ptr(), wb, _lib.current_stream()), "rp_sim_topk")

        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        out[f"{n}_gpus_{BQ}q_x_{rows}rows"] = e0.elapsed_time(e1) / 20 * 1e3
        if n > 1:
            # ... and the merge behind the exchange at that world size, on this one GPU: n sorted lists (the shard call's own
            # output, replicated with disjoint id ranges) of a rank's OWN 256 queries -> rp_topk_merge_strided, as the step runs it
            blk = B_STATES * (2 * TOP_K + 1)
            recv = torch.empty((n, blk), dtype=torch.int32, device=dev)
            g_s = recv[:, : B_STATES * TOP_K].view(torch.float32).view(n, B_STATES, TOP_K)
            g_i = recv[:, B_STATES * TOP_K : 2 * B_STATES * TOP_K].view(n, B_STATES, TOP_K)
            g_c = recv[:, 2 * B_STATES * TOP_K :]
            for r in range(n):
                g_s[r].copy_(o_s[:B_STATES])
                g_i[r].copy_(o_i[:B_STATES] + r * rows)
                g_c[r].copy_(o_c[:B_STATES])
            mwb = lib.rp_topk_merge_workspace_bytes(n, B_STATES, TOP_K)
            mws = torch.empty(mwb, dtype=torch.uint8, device=dev)
            f_s = torch.empty((B_STATES, TOP_K), dtype=torch.float32, device=dev)
            f_i = torch.empty((B_STATES, TOP_K), dtype=torch.int32, device=dev)
            f_c = torch.empty((B_STATES,), dtype=torch.int32, device=dev)

            def merge():
                _lib.check(lib.rp_topk_merge_strided(g_s.data_ptr(), g_i.data_ptr(), g_c.data_ptr(), g_s.stride(0), n, B_STATES, TOP_K,
                                                     f_s.data_ptr(), f_i.data_ptr(), f_c.data_ptr(), mws.data_ptr(), mwb,
                                                     _lib.current_stream()), "rp_topk_merge_strided")

            for _ in range(3):
                merge()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                merge()
            e1.record()
            torch.cuda.synchronize()
            out[f"{n}_gpus_merge_of_{n}_lists_us"] = e0.elapsed_time(e1) / 20 * 1e3
    return out


def train_step_leg(cfg, sd_dev, dev, batch_size, num_negatives=3, max_seq_len=1024, steps=5, dropout_rate=0.1):
    """One training step of the retriever as retrieval/confs/cli_lean4_random.yaml runs it (batch_size 8 per GPU, 3
    negatives, max_seq_len 1024, AdamW + gradient_clip_val 1.0): forward over the batch's contexts, positives and
    negatives as ONE packed pass (T5's dropout 0.1 at its six sites, as the reference's training mode has it), contrastive
    MSE and its backward, encoder backward, gradient norm, clipped AdamW over the flat buffers, refresh of the bf16 compute
    copies.  Contexts draw their lengths from the state mix, premises from
    the premise mix (both clipped to max_seq_len).  Wall time per step over `steps` steps (synchronised), then the
    per-class kernel times of an instrumented repeat; GEMM FLOPs: forward linear 2 T P, backward dgrad + wgrad 4 T P
    (P = the encoder's 217 M linear parameters, T = the pass's padded token count)."""
    from reprover_amd.train import HipT5Trainer, contrastive_mse, contrastive_mse_backward

    rng = np.random.default_rng(synth.SEED + 900 + batch_size)
    n_seq = batch_size * (2 + num_negatives)
    lens = np.concatenate([np.minimum(synth.synth_lengths(rng, batch_size, "mix", lo=16, hi=2048), max_seq_len),
                           np.minimum(synth.synth_lengths(rng, n_seq - batch_size, "mix", lo=8, hi=2048), max_seq_len)])
    ids, cu = synth.synth_token_batch(rng, lens)
    label = torch.zeros(batch_size, batch_size * (1 + num_negatives), device=dev)
    label[torch.arange(batch_size), torch.arange(batch_size)] = 1.0
    tr = HipT5Trainer(cfg, {k: v for k, v in sd_dev.items()}, dev, lr=1e-4, warmup_steps=0, gradient_clip_val=1.0,
                      dropout_rate=dropout_rate)

    def one_step():
        emb = tr.forward(ids, cu)
        ctx, prem = emb[:batch_size], emb[batch_size:]
        loss, sim, lab = contrastive_mse(ctx, prem, label)
        d_emb = torch.empty_like(emb)
        contrastive_mse_backward(ctx, prem, sim, lab, d_emb[:batch_size], d_emb[batch_size:])
        tr.backward(d_emb)
        tr.optimizer_step()
        return loss

    l0 = float(one_step())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one_step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    _lib.profile_enable(True)
    for _ in range(steps):
        one_step()
    torch.cuda.synchronize()
    prof = {k: v[0] / steps for k, v in _lib.profile_read().items()}
    _lib.profile_enable(False)
    T = int(cu[-1])
    Tp = (T + 255) // 256 * 256
    D, F_, inner, L = cfg["d_model"], cfg["d_ff"], cfg["num_heads"] * cfg["d_kv"], cfg["num_layers"]
    lin = 2.0 * Tp * L * (D * 3 * inner + inner * D + D * 2 * F_ + F_ * D)  # forward linear FLOPs of the pass
    fwd_gemm_ms = sum(prof[k] for k in ("gemm_qkv", "gemm_o", "gemm_wi", "gemm_wo"))
    tf = lambda flops, ms_: flops / (ms_ * 1e-3) / 1e12 if ms_ > 0 else 0.0
    out = {
        "ms_per_step": ms, "sequences": int(n_seq), "tokens": T, "tokens_per_s": T / (ms * 1e-3),
        "examples_per_s": batch_size / (ms * 1e-3), "loss_first_last": [l0, float(loss)],
        "kernel_ms_per_step": prof,
        "forward_gemm_tflops": tf(lin, fwd_gemm_ms), "dgrad_tflops": tf(lin, prof["bwd_dgrad"]),
        "wgrad_tflops": tf(lin, prof["bwd_wgrad"]),
        "backward_gemm_tflops": tf(2.0 * lin, prof["bwd_dgrad"] + prof["bwd_wgrad"]),
        "backward_gemm_mfma_frac": tf(2.0 * lin, prof["bwd_dgrad"] + prof["bwd_wgrad"]) / PEAK_BF16_TFLOPS,
        "whole_step_mfma_frac": tf(3.0 * lin, ms) / PEAK_BF16_TFLOPS,
        "config": f"batch_size {batch_size}, {num_negatives} negatives, max_seq_len {max_seq_len}, AdamW lr 1e-4, "
                  f"gradient_clip_val 1.0, dropout {dropout_rate} (T5's dropout_rate: the reference's training mode); one packed "
                  f"pass of {n_seq} sequences",
    }
    del tr
    torch.cuda.empty_cache()
    return out


def kernel_source_hash() -> str:
    """sha256 (16 hex digits) over the HIP sources: stamps profiles/pmc_traffic.json so that a counter-derived
    traffic figure is only quoted for the kernels it was measured on."""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "reprover_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if os.path.isdir(os.path.join(csrc, name)):  # csrc/probes/: bodies of probe builds, never part of the product library
            continue
        with open(os.path.join(csrc, name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


class _Leg:
    """``with _Leg(name, errors):`` - an auxiliary measurement whose failure must not cost the headline line: the exception
    is recorded under ``leg_errors[name]`` in the JSON and printed on stderr, and the run goes on."""

    def __init__(self, name, errors):
        self.name, self.errors = name, errors

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is None or not issubclass(et, Exception):
            return False
        import traceback

        self.errors[self.name] = f"{et.__name__}: {ev}"
        print(f"[bench] leg {self.name} failed:", file=sys.stderr)
        traceback.print_exception(et, ev, tb, file=sys.stderr)
        return True


def run_c5(args):
    """`--config c5`: BASELINE.json configs[4] at FULL size on one MI355X - ByT5-base encoder (random init, 18 layers) over
    256 states of the benchmark's length mix + masked top-100 over 1,000,000 x 1536 synthetic premises held as an e4m3
    index (`rp_sim_topk_fp8`; the bf16-index scan of the same rows is reported beside it).  A supplementary line in the
    contract's shape (the contract bench is configs[1]); the 1 M-row index is 1.5 GB of codes + 3 GB of bf16 rows: N = 1 only."""
    assert args.gpus == 1, "--config c5 is a single-GPU line (configs[4] names no sharding)"
    from reprover_amd.common import Fp8Index

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    build.build()
    lib = _lib.load()
    cfg = synth.t5_config("byt5-base")
    D = cfg["d_model"]
    enc = HipT5Encoder(cfg, random_init_state_dict(cfg, dev, 1), dev, torch.bfloat16)
    N, B, k, F = 1_000_000, B_STATES, TOP_K, N_FILES
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    codes = torch.empty((N, D), dtype=torch.uint8, device=dev)
    scale = torch.empty((N,), dtype=torch.float32, device=dev)
    E16 = torch.empty((N, D), dtype=torch.bfloat16, device=dev)
    for lo in range(0, N, 125_000):
        x = torch.nn.functional.normalize(torch.randn(125_000, D, generator=g, device=dev), dim=1)
        E16[lo : lo + 125_000] = x.to(torch.bfloat16)
        q8 = Fp8Index.quantize(x)
        codes[lo : lo + 125_000], scale[lo : lo + 125_000] = q8.codes, q8.scale
    rng = np.random.default_rng(5)
    lens = synth.synth_lengths(rng, B, "mix", lo=16, hi=2048)
    from reprover_amd import tokenizer

    ids_np, cu_np = tokenizer.encode_packed([synth.synth_state(rng, int(n) - 1) for n in lens], 2048)
    T, max_len = int(cu_np[-1]), int(np.diff(cu_np).max())
    ids_d, cu_d = torch.from_numpy(ids_np).to(dev), torch.from_numpy(cu_np).to(dev)
    (file_of, end_key, bits_t, own, qk), _ = synth.synth_masks(rng, N, B, F)
    f_d, ek_d = torch.from_numpy(file_of).to(dev), torch.from_numpy(end_key).to(dev)
    bt_d = torch.from_numpy(bits_t.view(np.int32)).to(dev)
    own_d, qk_d = torch.from_numpy(own).to(dev), torch.from_numpy(qk).to(dev)
    q = torch.empty((B, D), dtype=torch.bfloat16, device=dev)
    out_s = torch.empty((B, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((B, k), dtype=torch.int32, device=dev)
    out_c = torch.empty((B,), dtype=torch.int32, device=dev)
    nb = lib.rp_sim_topk_workspace_bytes(B, N, D, k, 0)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)

    def step(fp8=True):
        enc.encode_packed_device(ids_d, cu_d, B, T, max_len, q)
        if fp8:
            q8 = Fp8Index.quantize(q)  # the queries' e4m3 form: part of the step
            _lib.check(lib.rp_sim_topk_fp8(q8.codes.data_ptr(), q8.scale.data_ptr(), codes.data_ptr(), scale.data_ptr(), B, N, D,
                                           f_d.data_ptr(), ek_d.data_ptr(), bt_d.data_ptr(), F, own_d.data_ptr(), qk_d.data_ptr(),
                                           0, k, 0, out_s.data_ptr(), out_i.data_ptr(), out_c.data_ptr(), ws.data_ptr(), nb,
                                           _lib.current_stream()), "rp_sim_topk_fp8")
        else:
            _lib.check(lib.rp_sim_topk(q.data_ptr(), E16.data_ptr(), B, N, D, f_d.data_ptr(), ek_d.data_ptr(), bt_d.data_ptr(), F,
                                       own_d.data_ptr(), qk_d.data_ptr(), 0, k, 0, out_s.data_ptr(), out_i.data_ptr(),
                                       out_c.data_ptr(), ws.data_ptr(), nb, _lib.current_stream()), "rp_sim_topk")

    res = {}
    Tp = (T + 255) // 256 * 256
    for name, fp8 in (("e4m3_index", True), ("bf16_index", False)):
        for _ in range(max(args.warmup, 1)):
            step(fp8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(fp8)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        _lib.profile_enable(True)
        for _ in range(args.steps):
            step(fp8)
        torch.cuda.synchronize()
        prof = _lib.profile_read()
        _lib.profile_enable(False)
        wi_ms, wi_n = prof["gemm_wi"]
        scan_ms = (prof["scan"][0] + prof["scan_sample"][0]) / args.steps
        whole_ms = sum(prof[c][0] for c in ("scan", "scan_sample", "select")) / args.steps
        row_bytes = D * (1 if fp8 else 2) + (4 if fp8 else 0)
        scan_bytes = N * row_bytes + N * 12 + B * (row_bytes + 12) + B * k * 8
        wi_flops = 2.0 * Tp * D * 2 * cfg["d_ff"]
        res[name] = {"ms_per_step": dt * 1e3, "queries_per_s": B / dt, "counts_eq_k": bool((out_c == k).all()),
                     "scan_kernels_ms": scan_ms, "whole_rp_sim_topk_ms": whole_ms, "scan_bytes": int(scan_bytes),
                     "scan_hbm_gbs": scan_bytes / (scan_ms * 1e-3) / 1e9, "scan_hbm_frac": scan_bytes / (scan_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                     "scan_mfma_tflops": 2.0 * B * N * D / (scan_ms * 1e-3) / 1e12,
                     "ffn_in_gemm_tflops": wi_flops * wi_n / (wi_ms * 1e-3) / 1e12, "ffn_in_avg_launch_ms": wi_ms / max(wi_n, 1),
                     "kernel_ms_per_step": {c: v[0] / args.steps for c, v in prof.items()}}
    e = res["e4m3_index"]
    print(json.dumps({
        "metric": "retrieve QPS@top-100 (state encode + masked e4m3 similarity top-k), ByT5-base, 1M-premise corpus",
        "value": e["queries_per_s"], "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 1),
        "ms_per_step": e["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic",
        "config": {"workload": "configs[4]: ByT5-base encoder + fp8 (e4m3) similarity GEMM on MFMA, 1M synthetic premises, batch=256 "
                               "states, top-100", "n_premises": N, "n_files": F, "states_per_gpu": B, "k": k,
                   "state_tokens_per_gpu": T, "index": "e4m3 codes [1M, 1536] + f32 row scales; encoder GEMMs in bf16",
                   "weights": "random-init ByT5-base (d_model 1536, 18 layers, 12 heads, d_ff 3968)",
                   "all_counts_eq_k": e["counts_eq_k"]},
        "roofline": {"kernel": "gemm_kernel<EpiGegluBf16> (FFN wi_0|wi_1 GEMM of ByT5-base + gated-GELU epilogue)", "bound": "mfma",
                     "achieved": e["ffn_in_gemm_tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": e["ffn_in_gemm_tflops"] / PEAK_BF16_TFLOPS, "traffic": None, "avg_launch_ms": e["ffn_in_avg_launch_ms"]},
        "roofline_scan": {"kernel": "sim_scan_kernel (sample) + sim_filter_kernel (e4m3, v_mfma_scale_f32_32x32x64_f8f6f4)", "bound": "hbm",
                          "achieved": e["scan_hbm_gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": e["scan_hbm_frac"],
                          "traffic": None, "bytes_per_step": e["scan_bytes"], "ms_per_step": e["scan_kernels_ms"],
                          "whole_call_ms": e["whole_rp_sim_topk_ms"], "mfma_tflops": e["scan_mfma_tflops"]},
        "e4m3_index": e, "bf16_index": res["bf16_index"],
    }), flush=True)


def run_plumbing_only(args, world, rank, local, exchange_how):
    """`--plumbing-only`: everything of the N-rank run that is not GPU work, on the host (gloo): the rendezvous from the
    launcher's environment, the one-builder-per-node barrier, this rank's row shard, the two collectives of a step on CPU
    blocks of the step's shapes (query embeddings; the packed [scores | ids | counts] block as all-gather and as sliced
    all-to-all, merged by a host merge and checked against the unsharded answer), the barrier-bracketed max-over-ranks clock,
    ONE JSON line from rank 0.  No kernel runs and nothing is measured: the line says so."""
    from reprover_amd.dist import sliced_exchange_merge

    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("gloo")
        if local == 0:
            build.build()  # (cross-compiles without a GPU; a no-op when the shipped .so is current)
        dist.barrier()
    else:
        build.build()
    _lib.load()
    N, D, k, Bq = 4096, 64, 10, 8  # small stand-ins: the plumbing does not depend on the sizes
    bounds = np.linspace(0, N, world + 1).astype(int)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    BQ = Bq * world
    g = torch.Generator().manual_seed(synth.SEED)
    E = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=1)
    q_all_ref = torch.nn.functional.normalize(torch.randn(BQ, D, generator=g), dim=1)
    q_loc = q_all_ref[rank * Bq : (rank + 1) * Bq].contiguous()
    q_all = torch.empty((BQ, D))
    n_coll = 0

    def gather(dst, src):
        nonlocal n_coll
        n_coll += 1
        if world > 1:
            dist.all_gather(list(dst.view((world,) + tuple(src.shape)).unbind(0)), src)
        else:
            dst.copy_(src.view_as(dst))

    def host_merge(g_ids, g_scores, g_counts):  # [world, B, k] lists -> top-k per query (ties to the lower id)
        W, B, _ = g_scores.shape
        s = g_scores.permute(1, 0, 2).reshape(B, W * k)
        i = g_ids.permute(1, 0, 2).reshape(B, W * k)
        key = torch.argsort(-s.double() + i.double() * 1e-12, dim=1)[:, :k]
        return torch.gather(i, 1, key), torch.gather(s, 1, key), g_counts.sum(0).clamp(max=k)

    t0 = time.perf_counter()
    gather(q_all, q_loc)  # collective 1: query embeddings
    S = q_all @ E[lo:hi].T
    sc, ix = torch.topk(S, k, dim=1)
    ids = (ix + lo).to(torch.int32)
    cnt = torch.full((BQ,), k, dtype=torch.int32)
    blk = BQ * (2 * k + 1)
    send = torch.cat([sc.contiguous().view(torch.int32).reshape(-1), ids.reshape(-1), cnt])
    recv = torch.empty((world, blk), dtype=torch.int32)
    gather(recv, send)  # collective 2, all-gather form: the packed block
    g_s = recv[:, : BQ * k].view(torch.float32).view(world, BQ, k)
    g_i = recv[:, BQ * k : 2 * BQ * k].view(world, BQ, k)
    g_c = recv[:, 2 * BQ * k :]
    mine = slice(rank * Bq, (rank + 1) * Bq)
    ag = host_merge(g_i[:, mine], g_s[:, mine], g_c[:, mine])
    a2a = sliced_exchange_merge(ids, sc, cnt, None, merge=host_merge) if world > 1 else ag
    want_s, want_i = torch.topk(q_loc @ E.T, k, dim=1)
    ok = bool(torch.equal(ag[0], want_i.to(torch.int32)) and torch.equal(a2a[0], want_i.to(torch.int32))
              and torch.allclose(ag[1], want_s) and torch.equal(q_all, q_all_ref))
    dt = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        tt = torch.tensor([dt, float(ok)], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dmin = torch.tensor([float(ok)], dtype=torch.float64)
        dist.all_reduce(dmin, op=dist.ReduceOp.MIN)
        dt, ok = float(tt[0]), bool(dmin[0] == 1.0)
    if rank == 0:
        print(json.dumps({"metric": "plumbing only (no GPU work, nothing measured)", "plumbing_only": True, "value": None,
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": "weak",
                          "config": {"exchange": exchange_how if world > 1 else None, "collectives_per_step": 2 if world > 1 else 0,
                                     "shard_rows": [int(b) for b in bounds], "backend": "gloo",
                                     "sharded_merge_equals_single_gpu": ok},
                          "max_over_ranks_s": dt}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=("c2", "c5"), default="c2",
                    help="c2 (default): BASELINE.json configs[1], the contract's headline workload; c5: configs[4] at full size "
                         "(ByT5-base + e4m3 similarity, 1M premises) as a supplementary one-GPU line")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", choices=("auto", "allgather", "alltoall"), default=os.environ.get("RP_BENCH_EXCHANGE", "auto"),
                    help="N > 1: how the per-shard top-k lists reach the rank that merges them: ONE packed all-gather "
                         "(north_star's form) or ONE all-to-all of per-destination slices (1 / N of the bytes: in this step a rank "
                         "merges only its OWN 256 queries).  auto (default): by bytes - the all-gather below 4 ranks, the "
                         "all-to-all from 4 ranks on (at N = 8 the all-gather delivers 11.5 MB per rank of which 1.4 MB are "
                         "merged); `topk_only` reports both forms either way, and the product's replicated-query predict keeps "
                         "the all-gather (every rank needs every list there)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="run ONLY the multi-rank plumbing of this script on the host - argument checks, the self-launch of the "
                         "N ranks, rendezvous, the one-builder-per-node barrier, shard bounds, one packed all-gather + one sliced "
                         "all-to-all of CPU blocks, the max-over-ranks clock - and print a line marked plumbing_only (no GPU, no "
                         "measurement; tests/test_dist_cpu.py drives it at N = 8 under gloo)")
    ap.add_argument("--premise-sample", type=int, default=4096, help="premises in the encode-throughput leg")
    ap.add_argument("--no-full-reindex", action="store_true", help="skip the 130,000-premise reindex_corpus leg (~20 s)")
    ap.add_argument("--no-train-step", action="store_true", help="skip the training-step leg")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the premise-encode and scan-only legs (used under rocprofv3 so that every launch of "
                         "the dominant kernel has the step's shape and the stats average is comparable)")
    args = ap.parse_args()
    if args.config == "c5":
        return run_c5(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the command
        # line the driver contract names) and hand their output through; rank 0 prints the JSON line.
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        sys.exit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {args.gpus})")
    if args.exchange == "auto":
        args.exchange = "alltoall" if world >= 4 else "allgather"
        exchange_how = f"{args.exchange} (auto: {'>= 4 ranks' if world >= 4 else '< 4 ranks'})"
    else:
        exchange_how = args.exchange
    if args.plumbing_only:
        return run_plumbing_only(args, world, rank, local, exchange_how)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # RP_BENCH_SHARE_GPU=1 + RP_BENCH_BACKEND=gloo: functional smoke run of the N > 1 code path on a
    # single-GPU box (every rank on device 0; not a measurement).
    share = os.environ.get("RP_BENCH_SHARE_GPU") == "1"
    local = 0 if share else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist

        backend = os.environ.get("RP_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if world > 1:  # one builder per node; the others wait (the .so normally ships prebuilt and is up to date)
        if local == 0 or share:
            build.build()
        dist.barrier()
    else:
        build.build()
    lib = _lib.load()

    cfg = synth.t5_config("byt5-small")
    D = cfg["d_model"]
    sd = random_init_state_dict(cfg, dev, seed=synth.SEED)
    enc = HipT5Encoder(cfg, sd, dev, torch.bfloat16)

    # ---- corpus: 130k premises in 5k files, identical on every rank; this rank's row shard ------
    tmp = tempfile.mkdtemp()
    corpus_path = os.path.join(tmp, "corpus.jsonl")
    synth.write_corpus_jsonl(corpus_path, fast_corpus_records(N_FILES, N_PREMISES, synth.SEED))
    corpus = Corpus(corpus_path)
    N = len(corpus)
    bounds = np.linspace(0, N, world + 1).astype(int)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    g = torch.Generator(device=dev)
    g.manual_seed(synth.SEED + 1)
    E_full = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1).to(torch.bfloat16)
    E = E_full[lo:hi].contiguous()
    file_of = torch.from_numpy(corpus.file_of[lo:hi].copy()).to(dev)
    end_key = torch.from_numpy(corpus.end_key[lo:hi].copy()).to(dev)

    # ---- states: 256 per rank, byte lengths from the log-normal mix, files from the later half ---
    BQ = B_STATES * world
    all_ctx, all_txt = [], []
    for r in range(world):
        rng = np.random.default_rng(synth.SEED + 100 + r)
        if r == 0:
            lens0 = lens = synth.synth_lengths(rng, B_STATES, "mix", lo=16, hi=2048)
        else:  # weak scaling = the SAME work per GPU: every rank's 256 states carry rank 0's byte lengths (in an order of
            lens = rng.permutation(lens0)  # their own, other bytes, other files); independent draws differ by +-7 % in tokens
        for j in range(B_STATES):
            f = int(rng.integers(N_FILES // 2, N_FILES))
            txt = synth.synth_state(rng, int(lens[j]) - 1)
            all_txt.append(txt)
            all_ctx.append(Context(f"M/F{f}.lean", f"thm{r}_{j}", Pos(int(rng.integers(1, 60)), 0), txt))
    from reprover_amd import tokenizer

    my_txt = all_txt[rank * B_STATES : (rank + 1) * B_STATES]
    ids_np, cu_np = tokenizer.encode_packed(my_txt, 2048)
    T, max_len = int(cu_np[-1]), int(np.diff(cu_np).max())
    ids_d, cu_d = torch.from_numpy(ids_np).to(dev), torch.from_numpy(cu_np).to(dev)
    # Per-batch accessibility operand, as the product builds it (Corpus.device_query_masks): the host holds own_file / q_key
    # (12 bytes per query, pinned); every STEP copies them to the device (asynchronous, on the launch stream) and builds the
    # [F, ceil(B/32)] bit matrix there from the resident import closure (rp_build_file_bits).  Both are inside the timed
    # region since round 6 (rounds 1-5 built the operand on the host before it).
    bits_host, own, qk = corpus.query_masks(all_ctx)  # (the host-built bits only CHECK the device-built ones below)
    own_pin, qk_pin = torch.from_numpy(own).pin_memory(), torch.from_numpy(qk).pin_memory()
    own_d, qk_d = torch.empty(BQ, dtype=torch.int32, device=dev), torch.empty(BQ, dtype=torch.int64, device=dev)
    bits_d = torch.empty((corpus.num_files, (BQ + 31) // 32), dtype=torch.int32, device=dev)
    reach_d = corpus.device_reach(dev)

    def build_masks():
        own_d.copy_(own_pin, non_blocking=True)
        qk_d.copy_(qk_pin, non_blocking=True)
        _lib.check(lib.rp_build_file_bits(reach_d.data_ptr(), corpus.num_files, own_d.data_ptr(), BQ, bits_d.data_ptr(),
                                          _lib.current_stream()), "rp_build_file_bits")

    build_masks()
    torch.cuda.synchronize()
    assert np.array_equal(bits_d.cpu().numpy().view(np.uint32), bits_host), "device-built accessibility bits != host-built"
    n_acc = np.array([int(corpus.accessible_mask(c.path, c.theorem_pos).sum()) for c in all_ctx[:8]])

    q_loc = torch.empty((B_STATES, D), dtype=torch.bfloat16, device=dev)
    q_all = torch.empty((BQ, D), dtype=torch.bfloat16, device=dev) if world > 1 else q_loc
    # this rank's lists as views of ONE packed block [scores | ids | counts]: what the all-gather sends (dist.py)
    from reprover_amd.dist import packed_topk_buffers

    out_i, out_s, out_c = packed_topk_buffers(BQ, TOP_K, dev)
    send_block = torch.empty(0, dtype=torch.int32, device=dev).set_(out_s.untyped_storage(), 0, (BQ * (2 * TOP_K + 1),))
    ws_bytes = lib.rp_sim_topk_workspace_bytes(BQ, hi - lo, D, TOP_K, 0)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    if world > 1:
        blk = BQ * (2 * TOP_K + 1)  # 4-byte units per rank
        g_all = torch.empty((world, blk), dtype=torch.int32, device=dev)  # the receive buffer, read in place by the merge
        g_s = g_all[:, : BQ * TOP_K].view(torch.float32)
        g_i = g_all[:, BQ * TOP_K : 2 * BQ * TOP_K]
        g_c = g_all[:, 2 * BQ * TOP_K :]
        mws_bytes = lib.rp_topk_merge_workspace_bytes(world, B_STATES, TOP_K)
        mws = torch.empty(mws_bytes, dtype=torch.uint8, device=dev)
        f_s = torch.empty((B_STATES, TOP_K), dtype=torch.float32, device=dev)
        f_i = torch.empty((B_STATES, TOP_K), dtype=torch.int32, device=dev)
        f_c = torch.empty((B_STATES,), dtype=torch.int32, device=dev)

    def scan():
        _lib.check(lib.rp_sim_topk(q_all.data_ptr(), E.data_ptr(), BQ, hi - lo, D, file_of.data_ptr(),
                                   end_key.data_ptr(), bits_d.data_ptr(), corpus.num_files, own_d.data_ptr(),
                                   qk_d.data_ptr(), lo, TOP_K, 0, out_s.data_ptr(), out_i.data_ptr(),
                                   out_c.data_ptr(), ws.data_ptr(), ws_bytes, _lib.current_stream()), "rp_sim_topk")

    n_collectives = [0]

    def gather(dst, src):  # one all-gather: RCCL ncclAllGather on GPUs
        n_collectives[0] += 1
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(dst, src)
        else:
            dist.all_gather(list(dst.view((world,) + tuple(src.shape)).unbind(0)), src)

    def merge_into_f(g_ids, g_scores, g_counts):  # strided merge of [world, B_STATES, k] views of a receive buffer into f_*
        _lib.check(lib.rp_topk_merge_strided(g_scores.data_ptr(), g_ids.data_ptr(), g_counts.data_ptr(), g_scores.stride(0), world,
                                             B_STATES, TOP_K, f_s.data_ptr(), f_i.data_ptr(), f_c.data_ptr(), mws.data_ptr(),
                                             mws_bytes, _lib.current_stream()), "rp_topk_merge_strided")
        return f_i, f_s, f_c

    a2a_staging = {}

    def exchange_allgather():
        gather(g_all, send_block)  # one packed block per rank; every rank receives every rank's lists for ALL queries
        q0 = rank * B_STATES       # this rank merges its own queries, straight from the receive buffer
        merge_into_f(g_i.view(world, BQ, TOP_K)[:, q0 : q0 + B_STATES], g_s.view(world, BQ, TOP_K)[:, q0 : q0 + B_STATES],
                     g_c[:, q0 : q0 + B_STATES])

    def exchange_alltoall():
        from reprover_amd.dist import sliced_exchange_merge

        n_collectives[0] += 1      # one all-to-all of per-destination slices: a rank receives only its own queries' lists
        sliced_exchange_merge(out_i, out_s, out_c, None, merge=merge_into_f, staging=a2a_staging)

    exchange = {"allgather": exchange_allgather, "alltoall": exchange_alltoall}

    def step(how=None):
        how = how or args.exchange
        enc.encode_packed_device(ids_d, cu_d, B_STATES, T, max_len, q_loc)
        if world > 1:
            gather(q_all, q_loc)
        build_masks()  # 12 bytes per query H2D + the bit matrix from the resident closure: the product's per-batch work
        scan()
        if world > 1:
            exchange[how]()  # the step's second and last collective

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    try:
        for _ in range(args.warmup):
            step()
        barrier()
    except RuntimeError as e:  # a backend without the sliced all-to-all: the all-gather (north_star's form) always exists
        if world > 1 and args.exchange == "alltoall":
            print(f"[bench] rank {rank}: all-to-all exchange failed ({e}); falling back to the packed all-gather", file=sys.stderr)
            args.exchange, exchange_how = "allgather", "allgather (the all-to-all form failed on this backend)"
            for _ in range(args.warmup):
                step("allgather")
            barrier()
        else:
            raise
    n_collectives[0] = 0
    step()  # (untimed) count the collectives one step issues: 2 at N > 1 (query embeddings, packed result block), 0 at N = 1
    collectives_per_step = n_collectives[0]
    barrier()
    t0 = time.perf_counter()  # ---- the timed region: exactly K steps, nothing but the hot path's own launches
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # second pass, instrumented: an event pair around every launch (rp_profile_*) for the per-kernel split
    _lib.profile_enable(True)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    if world > 1:
        dist.barrier()
    counts_ok = bool(((f_c if world > 1 else out_c).cpu() == TOP_K).all())
    merged_ok = None
    reference_lists = None
    if world > 1 and rank == 0:
        # shard + exchange + merge must equal the single-GPU answer: check rank 0's own queries against a
        # scan of the whole (unsharded) matrix
        fo, eo = torch.from_numpy(corpus.file_of).to(dev), torch.from_numpy(corpus.end_key).to(dev)
        b0, o0, k0 = corpus.query_masks(all_ctx[:B_STATES])
        b0d = torch.from_numpy(b0.view(np.int32)).to(dev)
        o0d, k0d = torch.from_numpy(o0).to(dev), torch.from_numpy(k0).to(dev)
        r_s = torch.empty((B_STATES, TOP_K), dtype=torch.float32, device=dev)
        r_i = torch.empty((B_STATES, TOP_K), dtype=torch.int32, device=dev)
        r_c = torch.empty((B_STATES,), dtype=torch.int32, device=dev)
        wb = lib.rp_sim_topk_workspace_bytes(B_STATES, N, D, TOP_K, 0)
        wsx = torch.empty(wb, dtype=torch.uint8, device=dev)
        q0 = q_all[:B_STATES].contiguous()
        _lib.check(lib.rp_sim_topk(q0.data_ptr(), E_full.data_ptr(), B_STATES, N, D, fo.data_ptr(), eo.data_ptr(),
                                   b0d.data_ptr(), corpus.num_files, o0d.data_ptr(), k0d.data_ptr(), 0, TOP_K, 0,
                                   r_s.data_ptr(), r_i.data_ptr(), r_c.data_ptr(), wsx.data_ptr(), wb,
                                   _lib.current_stream()), "rp_sim_topk")
        torch.cuda.synchronize()
        reference_lists = (r_i, r_s)
        merged_ok = bool(torch.equal(r_i, f_i) and torch.equal(r_s, f_s))

    # ---- N > 1: the top-k stage WITH its collective: pre-encoded queries -> rp_sim_topk on the shard ->
    # the exchange -> the merge, timed together under both exchange forms, the collective alone from events
    topk_only = None
    if world > 1:
        topk_only = {}
        blk_bytes = BQ * (2 * TOP_K + 1) * 4
        for how in ("allgather", "alltoall"):
            try:  # (a backend that lacks one form must not cost the line: every rank fails the same call together)
                for _ in range(3):
                    scan()
                    exchange[how]()
                barrier()
            except RuntimeError as e:
                topk_only[how] = {"error": f"{type(e).__name__}: {e}"[:300]}
                continue
            t0 = time.perf_counter()
            iters = 20
            for _ in range(iters):
                scan()
                exchange[how]()
            torch.cuda.synchronize()
            dist.barrier()
            tdt = time.perf_counter() - t0
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
            for a, b in ev:  # the exchange alone (collective + pack copies + merge) between events on the launch stream
                scan()
                a.record()
                exchange[how]()
                b.record()
            torch.cuda.synchronize()
            ex_us = float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3
            tt = torch.tensor([tdt, ex_us], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ok = None
            if rank == 0:
                ok = bool(torch.equal(reference_lists[0], f_i) and torch.equal(reference_lists[1], f_s))
            topk_only[how] = {
                "qps": BQ * iters / float(tt[0].item()), "ms_per_step": float(tt[0].item()) / iters * 1e3,
                "exchange_us": float(tt[1].item()),
                # bytes one rank sends / receives in the collective (ring all-gather: its own block to every peer)
                "collective_bytes_sent_per_rank": (world - 1) * (blk_bytes if how == "allgather" else blk_bytes // world),
                "collective_bytes_received_per_rank": (world - 1) * (blk_bytes if how == "allgather" else blk_bytes // world),
                "bytes_merged_per_rank": blk_bytes,  # world lists of this rank's own B_STATES queries
                "sharded_merge_equals_single_gpu": ok,
            }
        qe = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in qe:
            a.record()
            gather(q_all, q_loc)
            b.record()
        torch.cuda.synchronize()
        topk_only["query_embedding_allgather_us"] = float(np.median([a.elapsed_time(b) for a, b in qe])) * 1e3
        topk_only["path"] = ("per step: rp_sim_topk of all N*256 queries on this rank's shard -> the exchange -> "
                             "rp_topk_merge_strided of the rank's own 256 queries; qps = N*256 queries / max-over-ranks time; "
                             "exchange_us = collective + (all-to-all: three pack copies) + merge between HIP events")
    ms_per_step = dt / args.steps * 1e3
    qps = BQ * args.steps / dt

    # ---- per-kernel roofline figures from the HIP-event records of the timed region --------------
    Tp = (T + 255) // 256 * 256  # the engine pads the token dimension to the GEMM tile
    F_ = cfg["d_ff"]
    wi_ms, wi_n = prof["gemm_wi"]
    wi_flops = 2.0 * Tp * D * 2 * F_
    wi_tf = wi_flops * wi_n / (wi_ms * 1e-3) / 1e12 if wi_ms > 0 else 0.0
    scan_ms, scan_n = prof["scan"][0] + prof["scan_sample"][0], prof["scan"][1] + prof["scan_sample"][1]
    n_loc = hi - lo
    scan_bytes = n_loc * D * 2 + n_loc * 12 + BQ * D * 2 + BQ * 12 + BQ * TOP_K * 8
    scan_gbs = scan_bytes * args.steps / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    tot_gemm_ms = sum(prof[k][0] for k in ("gemm_qkv", "gemm_o", "gemm_wi", "gemm_wo"))
    lin_flops = 2.0 * Tp * (D * 3 * enc.cfg["num_heads"] * 64 + enc.cfg["num_heads"] * 64 * D + D * 2 * F_ + F_ * D)
    all_gemm_tf = lin_flops * cfg["num_layers"] * args.steps / (tot_gemm_ms * 1e-3) / 1e12 if tot_gemm_ms else 0.0

    # ---- the encode pass's memory-bound kernels against the HBM peak (north_star: >= 70 % on encode) --------
    steps_n = max(args.steps, 1)
    n_layers = cfg["num_layers"]
    inner = enc.cfg["num_heads"] * 64
    B_ = B_STATES
    # chunk rows of the pooling pass that travel through `partial` (64-token chunks; one-chunk sequences are finished in place)
    n_multi = int(sum(-(-int(n) // 64) for n in np.diff(cu_np) if n > 64))
    hbm_rows = [
        # (kernel, profile class, launches per step, algorithmic bytes per launch)
        # x = a bf16 plane + an int8 extension plane since round 5: 3 bytes per element read, 3 written
        ("embed_kernel (byte-id gather -> bf16 plane + int8 extension plane of x + row statistic)", "embed", 1,
         T * 4 + Tp * D * 3 + Tp * 4),
        ("pool_partial_kernel + pool_finish_kernel (final RMSNorm + masked mean + L2 normalise)", "pool", 1,
         T * D * 3 + T * 4 + 2 * n_multi * D * 4 + B_ * D * 2),
        ("gemm_kernel<EpiResid8>, attention-out projection (K = 384: read-modify-write of the 3-byte form of x)", "gemm_o", n_layers,
         Tp * D * 6 + Tp * inner * 2 + D * inner * 2 + Tp * ((D + 63) // 64) * 4),
    ]
    roofline_hbm = []
    for name, cls_, per_step, nbytes in hbm_rows:
        ms, n = prof[cls_]
        us = ms / max(steps_n * per_step, 1) * 1e3
        gbs = nbytes / (us * 1e-6) / 1e9 if us > 0 else 0.0
        roofline_hbm.append({"kernel": name, "bytes": int(nbytes), "us": us, "achieved_gbs": gbs, "frac": gbs / PEAK_HBM_GBS,
                             "launches_per_step": per_step})
    whole_scan_ms = sum(prof[k][0] for k in ("scan", "scan_sample", "select")) / steps_n
    roofline_b1 = None

    # ---- scan-only QPS and premise-encode throughput (reported beside the headline value) --------
    barrier()
    scan_only_qps = scan_only_qps_fp8 = prem_per_s = prem_tok_per_s = prem_per_s_host = shard_call_us = None
    if not args.headline_only:
        t0 = time.perf_counter()
        for _ in range(20):
            scan()
        torch.cuda.synchronize()
        scan_only_qps = BQ * 20 / (time.perf_counter() - t0)
        # the same scan over the e4m3 copy of this rank's shard (BASELINE.json configs[4] flavour of the index)
        from reprover_amd.common import Fp8Index

        e8, q8 = Fp8Index.quantize(E), Fp8Index.quantize(q_all)

        def scan8():
            _lib.check(lib.rp_sim_topk_fp8(q8.codes.data_ptr(), q8.scale.data_ptr(), e8.codes.data_ptr(),
                                           e8.scale.data_ptr(), BQ, hi - lo, D, file_of.data_ptr(), end_key.data_ptr(),
                                           bits_d.data_ptr(), corpus.num_files, own_d.data_ptr(), qk_d.data_ptr(), lo,
                                           TOP_K, 0, out_s.data_ptr(), out_i.data_ptr(), out_c.data_ptr(),
                                           ws.data_ptr(), ws_bytes, _lib.current_stream()), "rp_sim_topk_fp8")

        scan8()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            scan8()
        torch.cuda.synchronize()
        scan_only_qps_fp8 = BQ * 20 / (time.perf_counter() - t0)
        if world == 1:
            shard_call_us = shard_shape_call_us(lib, corpus, E_full, dev)
        rngp = np.random.default_rng(synth.SEED + 7)
        plens = synth.synth_lengths(rngp, args.premise_sample, "mix", lo=8, hi=2048)
        pids, pcu = synth.synth_token_batch(rngp, plens)
        pout = torch.empty((args.premise_sample, D), dtype=torch.bfloat16, device=dev)
        enc.encode_packed(pids, pcu, pout)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        enc.encode_packed(pids, pcu, pout)
        torch.cuda.synchronize()
        pdt = time.perf_counter() - t0
        prem_per_s = args.premise_sample / pdt
        prem_tok_per_s = float(pcu[-1]) / pdt
        # second figure: the same pass from Python strings, i.e. including the host-side
        # byte tokenisation, packing and the H2D copy of the ids
        from reprover_amd.retrieval.model import PremiseRetriever

        retr = PremiseRetriever(enc, max_seq_len=2048)
        ptexts = [synth.synth_text(rngp, int(n) - 1) for n in plens]
        retr.encode_texts(ptexts, out=pout)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        retr.encode_texts(ptexts, out=pout)
        torch.cuda.synchronize()
        prem_per_s_host = args.premise_sample / (time.perf_counter() - t0)
    if world > 1 and not args.headline_only:
        agg = torch.tensor([prem_per_s, scan_only_qps, scan_only_qps_fp8], dtype=torch.float64, device=dev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)  # re-index shards by rank: throughputs add
        prem_per_s_host *= float(agg[0].item()) / prem_per_s  # same shard-parallel scaling
        prem_per_s = float(agg[0].item())
        prem_tok_per_s *= world
        scan_only_qps = float(agg[1].item()) / world  # every rank scanned all queries on its shard
        scan_only_qps_fp8 = float(agg[2].item()) / world

    # ---- N = 1 only: the product API from strings, length tiers, full re-index, single-state latency --------
    product = tiers = reindex = b1 = None
    leg_errors = {}  # an auxiliary leg that fails is reported here (and on stderr); the headline line is still printed
    if world == 1 and not args.headline_only:
        from reprover_amd.retrieval.model import PremiseRetriever
        from reprover_amd.tokenizer import ByT5Tokenizer

        with _Leg("product_api", leg_errors):
            retr = PremiseRetriever(enc, max_seq_len=1024, num_retrieved=TOP_K)  # predict conf: max_seq_len 1024
            retr.corpus, retr.corpus_embeddings, retr.embeddings_staled = corpus, E_full, False
            tok = ByT5Tokenizer()

            def predict_all(rounds=1, bs=64):  # datamodule.py:130-144 collate + model.py:281-327 predict_step, eval batch size 64
                retr.predict_step_outputs = []
                for i in [j for _ in range(rounds) for j in range(0, B_STATES, bs)]:
                    ctxs = all_ctx[i : i + bs]
                    t = tok([c.serialize() for c in ctxs], padding="longest", max_length=1024, truncation=True,
                            return_tensors="pt")
                    b = {"context": ctxs, "context_ids": t.input_ids, "context_mask": t.attention_mask}
                    for key in ("url", "commit", "file_path", "full_name", "start", "tactic_idx", "all_pos_premises"):
                        b[key] = [None] * len(ctxs)
                    retr.predict_step(b, 0)
                return retr.predict_step_outputs

            predict_all()
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                outs = predict_all()
                ts.append(time.perf_counter() - t0)
            ts4 = []  # the same states four times over = 16 batches: the fill (first collates) and the drain (last pass's
            for _ in range(3):  # records) of the pipeline weigh 1/4 as much as in the 4-batch figure, as in a real predict run
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                outs4 = predict_all(4)
                ts4.append(time.perf_counter() - t0)
            assert len(outs4) == 4 * B_STATES
            default_coalesce, retr.predict_coalesce_states = retr.predict_coalesce_states, 0
            predict_all()
            ts_nc = []  # one GPU pass per 64-state batch (predict_coalesce_states = 0): what the pass size costs
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                predict_all(4)
                ts_nc.append(time.perf_counter() - t0)
            retr.predict_coalesce_states = default_coalesce
            predict_all(1, B_STATES)
            ts256 = []  # eval_batch_size 256 (a data-module setting): the headline step's pass size through the product API - a
            for _ in range(3):  # 64-state pass costs 7 % more GPU time per token than a 256-state one (tools/padded_vs_packed.py)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                predict_all(4, B_STATES)
                ts256.append(time.perf_counter() - t0)
            product = {"qps": 4 * B_STATES / float(np.median(ts4)), "ms_per_256_states": float(np.median(ts4)) * 1e3 / 4,
                       "qps_4_batches": B_STATES / float(np.median(ts)),
                       "qps_16_batches_one_pass_per_batch": 4 * B_STATES / float(np.median(ts_nc)),
                       "qps_eval_batch_256": 4 * B_STATES / float(np.median(ts256)),
                       "predict_coalesce_states": default_coalesce,
                       "path": "strings -> ByT5 tokenizer (padding=longest, max_length 1024) -> predict_step at the "
                               "reference's eval batch of 64 states (host batches gathered into passes of "
                               "predict_coalesce_states states: packed encode + Corpus.get_nearest_premises incl. masks "
                               "from the device-resident closure, H2D/D2H, Premise mapping); qps = 16 batches (1024 states) "
                               "start to finish; qps_4_batches = 256 states, i.e. ONE pass whose collate and record "
                               "mapping overlap nothing",
                       "n_outputs": len(outs), "premises_per_output": len(outs[0]["retrieved_premises"])}

        with _Leg("length_tiers", leg_errors):
            tiers = {}
            for L, n_p in ((128, 2048), (512, 512), (2048, 128)):  # fixed byte-length tiers incl. EOS
                rngt = np.random.default_rng(synth.SEED + L)
                tids, tcu = synth.synth_token_batch(rngt, np.full(n_p, L))
                tout = torch.empty((n_p, D), dtype=torch.bfloat16, device=dev)
                enc.encode_packed(tids, tcu, tout)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                enc.encode_packed(tids, tcu, tout)
                torch.cuda.synchronize()
                tdt = time.perf_counter() - t0
                tiers[str(L)] = {"premises_per_s": n_p / tdt, "tokens_per_s": n_p * L / tdt, "premises": n_p}

        with _Leg("b1_latency", leg_errors):
            lat = {}
            for nbytes in (100, 300, 1000):  # the prover's call: one state per search node (tactic_generator.py:286-292)
                rngl = np.random.default_rng(synth.SEED + nbytes)
                st = synth.synth_state(rngl, nbytes)
                c0 = all_ctx[0]
                retr.retrieve(st, c0.path, c0.theorem_full_name, c0.theorem_pos, TOP_K)
                ls = []
                for _ in range(20):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    retr.retrieve(st, c0.path, c0.theorem_full_name, c0.theorem_pos, TOP_K)
                    ls.append(time.perf_counter() - t0)
                lat[str(nbytes)] = float(np.median(ls)) * 1e3
            b1 = {"retrieve_wall_ms_by_state_bytes": lat, "k": TOP_K,
                  "path": "PremiseRetriever.retrieve(): tokenise, encode, masked top-100 over the 130k index, D2H, Premise "
                          "objects; median of 20 calls"}
            # weight-streaming roofline of the single-state call: every call walks all encoder weights
            # (bf16, nothing survives in cache from call to call: 434 MB > the 256 MB Infinity Cache) and the bf16 index once
            w_bytes = n_layers * 2 * (3 * inner * D + D * inner + 2 * F_ * D + F_ * D) + cfg["vocab_size"] * D * 4
            idx_bytes = N * D * 2 + N * 12
            act_bytes = 101 * (n_layers * (2 * D * 8 + 3 * inner * 2 + inner * 2 + F_ * 2) + D * 4)
            b1_bytes = w_bytes + idx_bytes + act_bytes
            wall = lat["100"] * 1e-3
            roofline_b1 = {"kernel": "one retrieve() of a 100-byte state (hipGraph replay: 71 launches)", "bound": "hbm",
                           "bytes": int(b1_bytes), "weights_bytes": int(w_bytes), "index_bytes": int(idx_bytes),
                           "activation_bytes": int(act_bytes), "wall_ms": lat["100"],
                           "achieved": b1_bytes / wall / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "frac": b1_bytes / wall / 1e9 / PEAK_HBM_GBS}

        with _Leg("reindex_130k", leg_errors):
            if not args.no_full_reindex:
                tpath = os.path.join(tmp, "corpus_text.jsonl")
                synth.write_corpus_jsonl(tpath, fast_text_corpus_records(N_FILES, N_PREMISES, synth.SEED))
                t0 = time.perf_counter()
                r2 = PremiseRetriever(enc, max_seq_len=2048)
                r2.load_corpus(tpath)
                t_load = time.perf_counter() - t0
                t0 = time.perf_counter()
                r2.reindex_corpus(batch_size=64)
                torch.cuda.synchronize()
                t_idx = time.perf_counter() - t0
                n_tok = int(sum(min(len(pr.serialize().encode()) + 1, 2048) for pr in r2.corpus.all_premises[:2000]))
                # the same sweep through the index CLI (retrieval/index.py:13-41): checkpoint load, corpus load, re-index,
                # D2H and persist, as a user runs it (native index directory; BASELINE configs[3] at N = 1)
                from reprover_amd.retrieval import index as index_cli

                ckpt = os.path.join(tmp, "ckpt")
                enc.save_pretrained(ckpt)
                del r2.corpus_embeddings
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                index_cli.main(["--ckpt_path", ckpt, "--corpus-path", tpath, "--output-path", os.path.join(tmp, "index.rpidx/"),
                                "--batch-size", "64"])
                torch.cuda.synchronize()
                t_cli = time.perf_counter() - t0
                reindex = {"premises": len(r2.corpus), "reindex_corpus_s": t_idx, "premises_per_s": len(r2.corpus) / t_idx,
                           "index_cli_wall_s": t_cli, "index_cli_premises_per_s": len(r2.corpus) / t_cli,
                           "index_cli": "python -m reprover_amd.retrieval.index --ckpt_path .. --corpus-path .. --output-path "
                                        "index.rpidx/ (in process): HF checkpoint load + pack, corpus.jsonl load, re-index, "
                                        "persist (bf16 safetensors + closure arrays)",
                           "load_corpus_jsonl_s": t_load, "mean_tokens_per_premise_first_2000": n_tok / 2000.0,
                           "path": "PremiseRetriever.reindex_corpus(64) from corpus.jsonl: serialize (regex) + tokenise on the "
                                   "host, packed varlen encode on the GPU; BASELINE configs[3] at N = 1"}
                del r2

    # ---- box calibration: what this box's matrix pipes sustain under load (boxes of one pool differ by several per cent) ----
    box = None
    if rank == 0:
        with _Leg("box_calibration", leg_errors):
            sink = torch.zeros(4, dtype=torch.float32, device=dev)
            waves_per_cu, mfmas = 8, 32768

            def probe():
                _lib.check(lib.rp_dbg_mfma_probe(waves_per_cu, mfmas, sink.data_ptr(), _lib.current_stream()), "rp_dbg_mfma_probe")

            probe()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                probe()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            flops = 256.0 * waves_per_cu * mfmas * 32768.0
            tf = flops / (float(np.median(ts)) * 1e-3) / 1e12
            # ... and its memory side: a 1-GiB device-to-device copy (read + write counted), median of 5
            src = torch.empty(1 << 29, dtype=torch.bfloat16, device=dev).normal_()
            dst = torch.empty_like(src)
            dst.copy_(src)
            torch.cuda.synchronize()
            cs = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dst.copy_(src)
                e1.record()
                torch.cuda.synchronize()
                cs.append(e0.elapsed_time(e1))
            copy_gbs = 2.0 * src.numel() * 2 / (float(np.median(cs)) * 1e-3) / 1e9
            del src, dst
            box = {"mfma_probe_tflops": tf, "mfma_probe_frac_of_peak": tf / PEAK_BF16_TFLOPS,
                   "hbm_copy_gbs": copy_gbs, "hbm_copy_frac_of_peak": copy_gbs / PEAK_HBM_GBS,
                   "step_gemm_tflops_over_probe": all_gemm_tf / tf if tf else None,
                   "what": "v_mfma_f32_32x32x16_bf16 from registers, 8 waves per CU, pseudo-random bf16 operands, no memory "
                           "traffic (rp_dbg_mfma_probe): the matrix-pipe rate THIS box sustains under load, median of 5 launches "
                           "of ~1 ms; hbm_copy_gbs = a 1-GiB device-to-device copy (torch, read + write bytes); boxes of one pool differ by "
                           "several per cent in either, and step times from different boxes compare through them"}

    # ---- N = 1 only: the training step at the reference's training configuration -------------
    train = None
    if world == 1 and not args.headline_only and not args.no_train_step:
        with _Leg("train_step", leg_errors):
            train = {}
            for name, bsz in (("reference_conf_batch8", 8), ("batch64", 64)):
                train[name] = train_step_leg(cfg, sd, dev, bsz)
            train["reference_conf_batch8"]["ms_per_step_without_dropout"] = train_step_leg(cfg, sd, dev, 8, dropout_rate=0.0)["ms_per_step"]

    result = {
        "metric": "retrieve QPS@top-100 (state encode + masked similarity top-k), ByT5-small, 130k-premise corpus",
        "value": qps,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": "configs[1]: ByT5-small encode+retrieve, 130k premises, batch=256 states/GPU, top-100, bf16",
            "n_premises": N, "n_files": N_FILES, "states_per_gpu": B_STATES, "k": TOP_K,
            "state_tokens_per_gpu": T, "state_len_mix": "clip(round(LogNormal(ln 180, 0.9)), 16, 2048) bytes",
            "index": "row-sharded %d-way, bf16 unit-norm random rows" % world,
            "weights": "random-init ByT5-small (d_model 1472, 12 layers, 6 heads, d_ff 3584)",
            "accessible_premises_first_queries": n_acc.tolist(), "all_counts_eq_k": counts_ok, "sharded_merge_equals_single_gpu": merged_ok,
            "collectives_per_step": collectives_per_step, "exchange": exchange_how if world > 1 else None,
        },
        "premises_per_s": prem_per_s,
        "premises_per_s_incl_host_tokenisation": prem_per_s_host,
        "premise_tokens_per_s": prem_tok_per_s,
        "premise_len_mix": "clip(round(LogNormal(ln 180, 0.9)), 8, 2048) tokens, %d premises/GPU" % args.premise_sample,
        "premises_per_s_by_length_tier": tiers,
        "product_api_qps": product["qps"] if product else None,
        "product_api": product,
        "reindex_130k": reindex,
        "b1_latency_ms": b1,
        "scan_only_qps": scan_only_qps,
        "scan_only_qps_e4m3_index": scan_only_qps_fp8,
        "topk_only": topk_only,
        "roofline": {
            "kernel": "gemm_kernel<EpiGegluBf16> (FFN wi_0|wi_1 GEMM + gated-GELU epilogue; 58% of encoder FLOPs)",
            "bound": "mfma", "achieved": wi_tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": wi_tf / PEAK_BF16_TFLOPS, "traffic": None,
            "flops_per_launch": wi_flops, "avg_launch_ms": wi_ms / max(wi_n, 1), "launches": wi_n,
        },
        "roofline_scan": {
            "kernel": "sim_scan_kernel (sample pass) + sim_filter_kernel (filter pass) of one rp_sim_topk", "bound": "hbm",
            "achieved": scan_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": scan_gbs / PEAK_HBM_GBS, "traffic": None,
            "bytes_per_step": scan_bytes, "ms_per_step": scan_ms / args.steps,
            # MFMA utilisation of the similarity GEMM (north_star): 2 B N D flops over the same kernel time
            "mfma_tflops": 2.0 * BQ * n_loc * D * args.steps / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0,
            "mfma_frac": (2.0 * BQ * n_loc * D * args.steps / (scan_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS) if scan_ms > 0 else 0.0,
            "select_ms_per_step": prof["select"][0] / args.steps,
            # the WHOLE rp_sim_topk call (sample + its select + filter + gather/select: four launches) over the same bytes
            "whole_call_ms": whole_scan_ms,
            "whole_call_frac": (scan_bytes / (whole_scan_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if whole_scan_ms else None,
            # the same call timed WITHOUT the per-launch event pairs (calls back to back between two events, the 1-GPU entry of
            # shard_call_us): what a caller waits for; the figure above carries ~3 us of event overhead per launch
            "whole_call_ms_back_to_back": (shard_call_us or {}).get("1_gpus_256q_x_%drows" % N, 0.0) * 1e-3 or None,
            "whole_call_frac_back_to_back": (scan_bytes / ((shard_call_us or {}).get("1_gpus_256q_x_%drows" % N) * 1e-6) / 1e9 / PEAK_HBM_GBS)
            if (shard_call_us or {}).get("1_gpus_256q_x_%drows" % N) else None,
        },
        "roofline_hbm": roofline_hbm,
        "roofline_b1": roofline_b1,
        "shard_call_us": shard_call_us,
        "train_step": train,
        "all_encoder_gemms_tflops": all_gemm_tf,
        "box_calibration": box,
        "kernel_ms_per_step": {k: v[0] / args.steps for k, v in prof.items()},
    }
    traffic_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(traffic_file):  # HBM-side bytes per launch from the rocprofv3 PMC passes of this command
        tj = json.load(open(traffic_file))
        if tj.get("kernel_source_hash") == kernel_source_hash():  # measured on exactly these kernels
            result["roofline"]["traffic"] = tj.get("gemm_wi_bytes_per_launch")
            result["roofline"]["traffic_source"] = tj.get("source")
            result["roofline_scan"]["traffic"] = tj.get("scan_bytes_per_step")
        else:
            result["roofline"]["traffic_source"] = ("profiles/pmc_traffic.json was measured on other kernel sources "
                                                    "(hash mismatch): traffic withheld; re-run tools/profile_round.sh")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        with _Leg("cpu_baseline", leg_errors):
            result["cpu_baseline"] = cpu_baseline(cfg, sd, corpus_path, E_full, all_txt, all_ctx)
    if leg_errors:
        result["leg_errors"] = leg_errors
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
