"""Index a corpus with the retriever — same CLI as the reference's ``retrieval/index.py``
(lean-dojo/ReProver retrieval/index.py:13-41): ``--ckpt_path --corpus-path --output-path
--batch-size``.  Output: a pickled ``IndexedCorpus(corpus, fp32 CPU embeddings [N, D])``.
"""
from __future__ import annotations

import argparse
import logging
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
    args = parser.parse_args(argv)
    logger.info(args)

    # The reference falls back to the CPU with a warning (index.py:27-30); this engine has no CPU
    # path and says so instead of silently computing something else.
    if not torch.cuda.is_available():
        raise RuntimeError("reprover_amd needs an MI355X (HIP) device; no CPU fallback exists")
    device = torch.device("cuda")
    model = PremiseRetriever.load_hf(args.ckpt_path, 2048, device)
    model.load_corpus(args.corpus_path)
    model.reindex_corpus(batch_size=args.batch_size)
    if args.output_path.endswith(("/", ".rpidx")):
        # native index directory: bf16 embeddings as safetensors + the corpus jsonl (no pickle)
        save_index(args.output_path, args.corpus_path, model.corpus_embeddings)
    else:  # the reference's format: pickled IndexedCorpus with fp32 CPU embeddings (index.py:37-40)
        with open(args.output_path, "wb") as oup:
            pickle.dump(IndexedCorpus(model.corpus, model.corpus_embeddings.to(torch.float32).cpu()), oup)
    logger.info(f"Indexed corpus saved to {args.output_path}")


if __name__ == "__main__":
    main()
