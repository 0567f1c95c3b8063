"""The N > 1 path on CPU: world_size 2, gloo.  The collective plumbing of reprover_amd.dist (shard
bounds, id offsets, all-gather of the per-rank top-k lists, merge) is the product code; the two
compute steps are injected from the oracle here because the HIP kernels need a GPU (their sharded
parity is checked on the GPU in tests/test_kernels_gpu.py::test_shard_merge_equals_single_shot)."""
import json
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import common_ref
from reprover_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from reprover_amd.common import Context, Corpus, Pos
from reprover_amd.dist import (IndexShard, gather_shards, shard_bounds, sharded_get_nearest_premises,
                                sharded_nearest_premise_ids, sliced_exchange_merge)


def test_shard_bounds_balance_tokens():
    rng = np.random.default_rng(0)
    w = rng.integers(8, 2048, size=10_000)
    for world in (1, 2, 3, 8):
        b = shard_bounds(w, world)
        assert b[0] == 0 and b[-1] == len(w) and np.all(np.diff(b) >= 0) and len(b) == world + 1
        loads = np.array([w[b[r] : b[r + 1]].sum() for r in range(world)])
        assert loads.max() - loads.min() <= 2 * 2048
    assert shard_bounds([5], 4).tolist()[-1] == 1


def _oracle_local_topk(shard, batch_context, query_emb, k):
    """Oracle stand-in for rp_sim_topk on this rank's rows (ids are global)."""
    E = shard.embeddings.numpy()
    S = query_emb.numpy() @ E.T
    acc = np.stack([shard.corpus.accessible_mask(c.path, c.theorem_pos)[shard.lo : shard.hi] for c in batch_context])
    B = len(batch_context)
    ids = np.full((B, k), -1, dtype=np.int32)
    sc = np.full((B, k), -np.inf, dtype=np.float32)
    cnt = np.zeros(B, dtype=np.int32)
    for j in range(B):
        n = int(min(k, acc[j].sum()))
        if n:
            li, ls = common_ref.masked_topk(S[j : j + 1], acc[j : j + 1], n)
            ids[j, :n] = li[0] + shard.lo
            sc[j, :n] = ls[0]
        cnt[j] = n
    return torch.from_numpy(ids), torch.from_numpy(sc), torch.from_numpy(cnt)


def _oracle_merge(g_ids, g_scores, g_counts):
    R, B, k = g_scores.shape
    ids = np.full((B, k), -1, dtype=np.int32)
    sc = np.full((B, k), -np.inf, dtype=np.float32)
    cnt = np.zeros(B, dtype=np.int32)
    for j in range(B):
        cand = [(-float(g_scores[r, j, i]), int(g_ids[r, j, i])) for r in range(R) for i in range(int(g_counts[r, j]))]
        cand.sort()
        n = min(k, len(cand))
        ids[j, :n] = [c[1] for c in cand[:n]]
        sc[j, :n] = [-c[0] for c in cand[:n]]
        cnt[j] = n
    return torch.from_numpy(ids), torch.from_numpy(sc), torch.from_numpy(cnt)


def _worker(rank, world, port, corpus_path, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        corpus = Corpus(corpus_path)
        N, D, B, k = len(corpus), 48, 12, 20
        rng = np.random.default_rng(123)  # same on every rank
        E = rng.standard_normal((N, D)).astype(np.float32)
        Q = rng.standard_normal((B, D)).astype(np.float32)
        files = [json.loads(l) for l in open(corpus_path)]
        ctxs = [Context(files[int(f)]["path"], f"t{j}", Pos(int(rng.integers(1, 300)), 0), "a ⊢ b")
                for j, f in enumerate(rng.integers(20, len(files), size=B))]
        weights = [len(p.code) for p in corpus.all_premises]
        shard = IndexShard(corpus, shard_bounds(weights, world), rank, torch.device("cpu"))
        shard.embeddings = torch.from_numpy(E[shard.lo : shard.hi].copy())
        # the step issues ONE collective (the packed [scores | ids | counts] block), whatever the backend entry point
        calls = []
        real_ag, real_agt = dist.all_gather, dist.all_gather_into_tensor
        dist.all_gather = lambda *a, **kw: (calls.append("all_gather"), real_ag(*a, **kw))[1]
        dist.all_gather_into_tensor = lambda *a, **kw: (calls.append("all_gather_into_tensor"), real_agt(*a, **kw))[1]
        try:
            ids, scores, counts = sharded_nearest_premise_ids(shard, ctxs, torch.from_numpy(Q), k, None,
                                                              local_topk=_oracle_local_topk, merge=_oracle_merge)
        finally:
            dist.all_gather, dist.all_gather_into_tensor = real_ag, real_agt
        assert len(calls) == 1, calls
        # single-process answer
        acc = np.stack([corpus.accessible_mask(c.path, c.theorem_pos) for c in ctxs])
        S = Q @ E.T
        for j in range(B):
            n = int(min(k, acc[j].sum()))
            assert counts[j] == n
            if n:
                wi, ws = common_ref.masked_topk(S[j : j + 1], acc[j : j + 1], n)
                assert ids[j, :n].tolist() == wi[0].tolist()
                assert np.allclose(scores[j, :n].numpy(), ws[0])
        # the sliced exchange (a rank owns queries [rank B/world, (rank + 1) B/world) and receives only their lists): ONE
        # all-to-all, no all-gather, and the same merged lists as the all-gather form for the rank's own queries
        l_ids, l_sc, l_cnt = _oracle_local_topk(shard, ctxs, torch.from_numpy(Q), k)
        calls2 = []
        real_a2a = dist.all_to_all_single
        dist.all_to_all_single = lambda *a, **kw: (calls2.append("all_to_all_single"), real_a2a(*a, **kw))[1]
        dist.all_gather = lambda *a, **kw: (calls2.append("all_gather"), real_ag(*a, **kw))[1]
        try:
            s_ids, s_sc, s_cnt = sliced_exchange_merge(l_ids, l_sc, l_cnt, None, merge=_oracle_merge)
        finally:
            dist.all_to_all_single, dist.all_gather = real_a2a, real_ag
        assert calls2 == ["all_to_all_single"], calls2
        Bq = B // world
        mine = slice(rank * Bq, (rank + 1) * Bq)
        assert torch.equal(s_ids, ids[mine]) and torch.equal(s_sc, scores[mine]) and torch.equal(s_cnt, counts[mine])
        # the drop-in wrapper raises ValueError exactly when the reference would
        short = [j for j in range(B) if acc[j].sum() < k]
        ok = [j for j in range(B) if acc[j].sum() >= k]
        if ok:
            prem, sc = sharded_get_nearest_premises(shard, [ctxs[j] for j in ok], torch.from_numpy(Q[ok]), k, None,
                                                    local_topk=_oracle_local_topk, merge=_oracle_merge)
            assert len(prem) == len(ok) and all(len(r) == k for r in prem)
            assert prem[0][0] is corpus.all_premises[int(ids[ok[0], 0])]
        raised = False
        try:
            sharded_get_nearest_premises(shard, ctxs, torch.from_numpy(Q), k, None,
                                         local_topk=_oracle_local_topk, merge=_oracle_merge)
        except ValueError:
            raised = True
        assert raised == bool(short)
        # multi-GPU re-index: uneven shards -> one padded all-gather -> the full matrix, rows in corpus order
        full = gather_shards(shard.embeddings, shard.bounds)
        assert full.shape == (N, D) and np.array_equal(full.numpy(), E)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write(f"{shard.lo} {shard.hi}")
    finally:
        dist.destroy_process_group()


def test_sharded_retrieval_world2_gloo():
    files = synth.synth_corpus_records(40, 600, seed=21, max_imports=6)
    d = tempfile.mkdtemp()
    path = os.path.join(d, "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    port = 29500 + os.getpid() % 2000
    mp.start_processes(_worker, args=(2, port, path, d), nprocs=2, join=True, start_method="spawn")
    spans = [tuple(map(int, open(os.path.join(d, f"ok{r}")).read().split())) for r in range(2)]
    assert spans[0][0] == 0 and spans[0][1] == spans[1][0] and spans[1][1] > spans[1][0]


def _run_bench(cmd, env):
    import subprocess

    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:]  # ONE line, from rank 0 only
    return json.loads(lines[0])


def test_bench_eight_rank_plumbing_under_gloo():
    """`bench.py --gpus 8` as the driver's scaling run starts it, minus the GPU work (`--plumbing-only`, gloo, CPU blocks):
    (a) the script launching its own 8 ranks, (b) the driver's launcher command line (`python -m torch.distributed.run
    --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...`) - rendezvous, the
    one-builder barrier, 8 row shards, the two collectives of a step in both exchange forms with the merge equal to the
    unsharded answer on every rank, the max-over-ranks clock, one JSON line; a --gpus / WORLD_SIZE mismatch is refused."""
    import socket
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(ROOT, "bench.py")
    d = _run_bench([sys.executable, bench, "--gpus", "8", "--steps", "3", "--warmup", "1", "--plumbing-only"], env)
    assert d["plumbing_only"] is True and d["n_gpus"] == 8 and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["sharded_merge_equals_single_gpu"] is True and d["config"]["collectives_per_step"] == 2
    assert d["config"]["exchange"].startswith("alltoall (auto")  # by bytes: all-to-all from 4 ranks on
    assert len(d["config"]["shard_rows"]) == 9 and d["config"]["shard_rows"][0] == 0
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    d = _run_bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
                    "127.0.0.1", "--master-port", str(port), bench, "--gpus", "8", "--steps", "2", "--warmup", "1",
                    "--plumbing-only", "--exchange", "allgather"], env)
    assert d["n_gpus"] == 8 and d["config"]["exchange"] == "allgather" and d["config"]["sharded_merge_equals_single_gpu"] is True
    d = _run_bench([sys.executable, bench, "--gpus", "2", "--plumbing-only"], env)
    assert d["config"]["exchange"].startswith("allgather (auto")
    bad = subprocess.run([sys.executable, bench, "--gpus", "4", "--plumbing-only"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in bad.stderr
