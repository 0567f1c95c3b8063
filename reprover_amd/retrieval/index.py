"""Index a corpus with the retriever — same CLI as the reference's ``retrieval/index.py``
(lean-dojo/ReProver retrieval/index.py:13-41): ``--ckpt_path --corpus-path --output-path
--batch-size``.  Output: a pickled ``IndexedCorpus(corpus, fp32 CPU embeddings [N, D])``.

Multi-GPU (BASELINE.json configs[3]; the reference indexes on one device): launch under
``python -m torch.distributed.run --nproc-per-node N -m reprover_amd.retrieval.index ...``; each rank
encodes the contiguous row range that holds 1/N of the corpus' tokens (no communication), one
all-gather assembles the matrix and rank 0 writes the same file a single-GPU run writes.
"""
from __future__ import annotations

import argparse
import logging
import os
import pickle

import torch

from ..common import IndexedCorpus, save_index
from .model import PremiseRetriever

logger = logging.getLogger(__name__)


def main(argv=None) -> None:
    parser = argparse.ArgumentParser(description="Index the premise corpus with the MI355X retriever.")
    parser.add_argument("--ckpt_path", type=str, required=True)
    parser.add_argument("--corpus-path", type=str, required=True)
    parser.add_argument("--output-path", type=str, required=True)
    parser.add_argument("--batch-size", type=int, default=64)
    parser.add_argument("--reference-pickle", action="store_true",
                        help="write the pickle under the REFERENCE's class names (common.IndexedCorpus, networkx closure, "
                             "lean_dojo Pos) so that the reference's prover / load_corpus can load it; needs networkx")
    args = parser.parse_args(argv)
    logger.info(args)

    # The reference falls back to the CPU with a warning (index.py:27-30); this engine has no CPU
    # path and says so instead of silently computing something else.
    if not torch.cuda.is_available():
        raise RuntimeError("reprover_amd needs an MI355X (HIP) device; no CPU fallback exists")
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # functional test of the N > 1 path on a one-GPU box: RP_DIST_SHARE_GPU=1 RP_DIST_BACKEND=gloo
    # (RCCL refuses two ranks on one device)
    if os.environ.get("RP_DIST_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    model = PremiseRetriever.load_hf(args.ckpt_path, 2048, device)
    model.load_corpus(args.corpus_path)
    if world > 1:
        import torch.distributed as dist

        from .. import dist as rdist

        backend = os.environ.get("RP_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        bounds = rdist.shard_bounds(rdist.premise_token_counts(model.corpus, model.max_seq_len), world)
        shard = rdist.IndexShard(model.corpus, bounds, rank, device)
        rdist.reindex_shard(model, shard)
        model.corpus_embeddings = rdist.gather_shards(shard.embeddings, bounds)
        model.embeddings_staled = False
        dist.barrier()
        if rank != 0:
            dist.destroy_process_group()
            return
    else:
        model.reindex_corpus(batch_size=args.batch_size)
    if args.output_path.endswith(("/", ".rpidx")):
        # native index directory: bf16 embeddings as safetensors + the corpus jsonl (no pickle)
        # + the closure bit rows / per-premise arrays the search consumes, and the e4m3 form when asked for
        fp8 = None
        if os.environ.get("RP_INDEX_FP8") == "1":
            from ..common import Fp8Index

            fp8 = Fp8Index.quantize(model.corpus_embeddings, device)
        save_index(args.output_path, args.corpus_path, model.corpus_embeddings, corpus=model.corpus, fp8=fp8)
    elif args.reference_pickle:  # ... under the reference's own class identities (prover/tactic_generator.py:273-276 loads it)
        from ..common import save_reference_pickle

        save_reference_pickle(args.output_path, model.corpus, model.corpus_embeddings)
    else:  # the reference's format: pickled IndexedCorpus with fp32 CPU embeddings (index.py:37-40)
        with open(args.output_path, "wb") as oup:
            pickle.dump(IndexedCorpus(model.corpus, model.corpus_embeddings.to(torch.float32).cpu()), oup)
    logger.info(f"Indexed corpus saved to {args.output_path}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
