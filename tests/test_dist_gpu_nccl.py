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


def test_sharded_step_over_rccl_world1():
    from reprover_amd import synth
    from reprover_amd.common import Context, Corpus, Pos
    from reprover_amd.dist import (IndexShard, launch_sharded_nearest_premises, shard_bounds, sharded_nearest_premise_ids)

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
    want = corpus.nearest_premise_ids(E, ctxs, Q, k)
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
    finally:
        dist.destroy_process_group()
