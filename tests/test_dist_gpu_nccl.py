"""RCCL on the device that is here: a world_size-1 `nccl` process group on the MI355X.  It cannot show scaling, but it runs
the sharded step's collective - `all_gather_into_tensor` of the packed [scores | ids | counts] block, an int32 storage alias
of three typed views - through the RCCL backend on device memory, followed by the strided merge, and requires the answer of the
unsharded search: dtype, contiguity and aliasing of what the product hands to RCCL are what the 8-GPU run will hand it."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _problem():
    from reprover_amd import synth
    from reprover_amd.common import Context, Corpus, Pos

    assert torch.cuda.is_available()
    files = synth.synth_corpus_records(60, 30000, seed=5, max_imports=6)
    path = os.path.join(tempfile.mkdtemp(), "corpus.jsonl")
    synth.write_corpus_jsonl(path, files)
    corpus = Corpus(path)
    N, D, B, k = len(corpus), 256, 40, 25
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    E = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device="cuda"), dim=1).to(torch.bfloat16)
    Q = torch.nn.functional.normalize(torch.randn(B, D, generator=g, device="cuda"), dim=1).to(torch.bfloat16)
    ctxs = [Context(files[40 + j % 20]["path"], f"t{j}", Pos(400, 0), "a ⊢ b") for j in range(B)]
    return corpus, E, Q, ctxs, k, corpus.nearest_premise_ids(E, ctxs, Q, k)


def test_sharded_step_over_rccl_world1():
    from reprover_amd.dist import (IndexShard, launch_sharded_nearest_premises, shard_bounds, sharded_nearest_premise_ids)

    corpus, E, Q, ctxs, k, want = _problem()
    N = len(corpus)
    port = 29700 + os.getpid() % 200
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        shard = IndexShard(corpus, shard_bounds(np.ones(N), 1), 0, torch.device("cuda", 0))
        shard.embeddings = E
        ids, scores, counts = sharded_nearest_premise_ids(shard, ctxs, Q, k)  # one all_gather_into_tensor + merge
        torch.cuda.synchronize()
        assert torch.equal(ids, want[0]) and torch.equal(scores, want[1]) and torch.equal(counts, want[2])
        prem, sc = launch_sharded_nearest_premises(shard, ctxs, Q, k).finish()  # the pipelined form predict_step uses
        assert [[p.full_name for p in row] for row in prem] == \
            [[corpus.all_premises[i].full_name for i in row] for row in want[0].cpu().tolist()]
        assert np.array_equal(np.array(sc, dtype=np.float32), want[1].cpu().numpy())
        # the library's own communicator seeded through this group (what RP_COMM=abi does in on_predict_start): RCCL
        # carries torch's communicator and the library's side by side
        from reprover_amd.dist import HipComm

        comm = HipComm.from_torch_group()
        try:
            assert (comm.rank, comm.world) == (0, 1)
            ids2, scores2, counts2 = sharded_nearest_premise_ids(shard, ctxs, Q, k, group=comm)
            torch.cuda.synchronize()
            assert torch.equal(ids2, want[0]) and torch.equal(scores2, want[1]) and torch.equal(counts2, want[2])
        finally:
            comm.close()
    finally:
        dist.destroy_process_group()


def test_sharded_step_over_the_abi_communicator_world1():
    """The same step with the collective behind the C ABI (rp_comm_unique_id / rp_comm_init / rp_allgather_topk /
    rp_comm_allgather: RCCL bound by the library itself, no torch.distributed anywhere): a world-size-1 communicator on
    the one GPU here.  Answers of the unsharded search, bit for bit; a second communicator can live beside the first."""
    from reprover_amd import _lib
    from reprover_amd.dist import (HipComm, IndexShard, all_gather_stack, gather_shards, launch_sharded_nearest_premises,
                                   shard_bounds, sharded_nearest_premise_ids)

    corpus, E, Q, ctxs, k, want = _problem()
    N = len(corpus)
    comm = HipComm(HipComm.unique_id(), 0, 1)
    try:
        lib = _lib.load()
        assert lib.rp_comm_world(comm._handle) == 1 and lib.rp_comm_rank(comm._handle) == 0
        x = torch.arange(3 * 1472, dtype=torch.float32, device="cuda").view(3, 1472).to(torch.bfloat16)
        assert torch.equal(all_gather_stack(x, comm), x[None])  # rp_comm_allgather: the query-embedding exchange
        assert torch.equal(gather_shards(E[:1000], np.array([0, 1000]), comm), E[:1000])
        shard = IndexShard(corpus, shard_bounds(np.ones(N), 1), 0, torch.device("cuda", 0))
        shard.embeddings = E
        ids, scores, counts = sharded_nearest_premise_ids(shard, ctxs, Q, k, group=comm)  # ONE rp_allgather_topk
        torch.cuda.synchronize()
        assert torch.equal(ids, want[0]) and torch.equal(scores, want[1]) and torch.equal(counts, want[2])
        prem, sc = launch_sharded_nearest_premises(shard, ctxs, Q, k, group=comm).finish()
        assert [[p.full_name for p in row] for row in prem] == \
            [[corpus.all_premises[i].full_name for i in row] for row in want[0].cpu().tolist()]
        assert np.array_equal(np.array(sc, dtype=np.float32), want[1].cpu().numpy())
        # a subset of the queries, as a rank of a larger world merges only its own: queries [8, 24)
        blk = torch.cat([want[1].view(torch.int32).reshape(-1), want[0].reshape(-1), want[2]])
        i2, s2, c2, recv = comm.allgather_topk(blk, len(ctxs), k, 8, 16)
        torch.cuda.synchronize()
        assert torch.equal(i2, want[0][8:24]) and torch.equal(s2, want[1][8:24]) and torch.equal(c2, want[2][8:24])
        assert torch.equal(recv[0], blk)
        other = HipComm(HipComm.unique_id(), 0, 1)
        assert torch.equal(other.all_gather_stack(x), x[None])
        other.close()
    finally:
        comm.close()
    # argument errors come back as statuses with a message, not as crashes
    import ctypes as C

    h = C.c_void_p()
    assert lib.rp_comm_init(C.c_char_p(b"\0" * 128), 3, 2, C.byref(h)) == -1
    assert b"rank 3 of 2" in lib.rp_last_error()
