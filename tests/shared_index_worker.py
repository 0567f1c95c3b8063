"""Worker process of tests/test_shared_index_gpu.py: attaches the owner's index and answers queries.
Run as a script:  python shared_index_worker.py <request.pickle> <response.pickle>"""
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(req_path: str, out_path: str) -> None:
    import torch

    from reprover_amd import synth
    from reprover_amd.common import Pos
    from reprover_amd.retrieval.model import PremiseRetriever

    req = pickle.load(open(req_path, "rb"))
    cfg = synth.t5_config(req["config"])
    model = PremiseRetriever.from_state_dict(cfg, synth.synth_state_dict(cfg), req["max_seq_len"], "cuda:0",
                                             index_dtype=req["index_dtype"])
    free_before = torch.cuda.mem_get_info()[0]
    model.attach_index(req["handle"], req["corpus_jsonl"])
    torch.cuda.synchronize()
    free_after = torch.cuda.mem_get_info()[0]
    same_memory = model.corpus_embeddings.data_ptr() != 0 and model.corpus_embeddings.is_cuda
    answers = []
    for use_graphs in (True, False):
        model.use_graphs = use_graphs
        for state, path, name, pos in req["queries"]:
            prem, sc = model.retrieve(state, path, name, Pos(*pos), req["k"])
            answers.append(([p.full_name for p in prem], sc))
    refused = False
    try:
        model.embeddings_staled = True
        model.reindex_corpus(8)
    except RuntimeError:
        refused = True
    # what the worker sees in the first rows of the shared matrix (the owner compares)
    head = model.corpus_embeddings[:4].float().cpu()
    pickle.dump({"answers": answers, "refused": refused, "head": head, "same_memory": same_memory,
                 "bytes_taken_by_attach": int(free_before - free_after)}, open(out_path, "wb"))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
