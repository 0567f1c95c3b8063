"""One index per GPU shared by worker processes (the reference replicates it per prover actor,
prover/proof_search.py:438-447): a second PROCESS attaches the owner's device-resident index through HIP IPC and
must return exactly the owner's answers - graph replay and launch by launch, bf16 and e4m3 - without taking a copy."""
import os
import pickle
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

from reprover_amd import synth
from reprover_amd.common import Pos
from reprover_amd.retrieval.model import PremiseRetriever

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("index_dtype", ["bf16", "fp8"])
def test_worker_process_searches_the_owners_index(index_dtype):
    cfg = synth.t5_config("tiny")
    n_files, n_prem = 40, 3000
    files = synth.synth_corpus_records(n_files, n_prem, seed=11)
    d = tempfile.mkdtemp()
    jsonl = os.path.join(d, "corpus.jsonl")
    synth.write_corpus_jsonl(jsonl, files)
    owner = PremiseRetriever.from_state_dict(cfg, synth.synth_state_dict(cfg), 512, "cuda:0", index_dtype=index_dtype)
    owner.load_corpus(jsonl)
    owner.reindex_corpus(64)
    rng = np.random.default_rng(2)
    last = owner.corpus.files[-1]
    queries = [(synth.synth_state(rng, n), last.path, f"t{n}", (10_000, 0)) for n in (20, 90, 200, 400)]
    k = 10
    handle = owner.share_index()
    pickle.dumps(handle)  # the handle travels by any byte channel
    req, out = os.path.join(d, "req.pickle"), os.path.join(d, "out.pickle")
    pickle.dump({"config": "tiny", "max_seq_len": 512, "index_dtype": index_dtype, "handle": handle,
                 "corpus_jsonl": jsonl, "queries": queries, "k": k}, open(req, "wb"))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "shared_index_worker.py"), req, out], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = pickle.load(open(out, "rb"))
    mine = []
    for use_graphs in (True, False):
        owner.use_graphs = use_graphs
        for state, path, name, pos in queries:
            prem, sc = owner.retrieve(state, path, name, Pos(*pos), k)
            mine.append(([p.full_name for p in prem], sc))
    owner.use_graphs = True
    assert res["answers"] == mine  # same kernels on the same memory: identical names and scores
    assert res["refused"], "an attached retriever must refuse to re-index"
    assert torch.equal(res["head"], owner.corpus_embeddings[:4].float().cpu())
    # attaching maps memory, it does not allocate a second matrix
    index_bytes = owner.corpus_embeddings.numel() * 2
    print(f"attach took {res['bytes_taken_by_attach']} B of device memory; the matrix is {index_bytes} B")
    assert res["bytes_taken_by_attach"] < index_bytes // 2 + (64 << 20)
