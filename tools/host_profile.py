"""Where does the host time of the product API go?  cProfile of predict_step (4 x 64 states) and of retrieve()
on the GPU box: python tools/host_profile.py"""
import cProfile, os, pstats, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from reprover_amd import synth
from reprover_amd.common import Context, Corpus, Pos
from reprover_amd.encoder import HipT5Encoder
from reprover_amd.retrieval.model import PremiseRetriever
from reprover_amd.tokenizer import ByT5Tokenizer

dev = torch.device("cuda:0")
cfg = synth.t5_config("byt5-small")
enc = HipT5Encoder(cfg, bench.random_init_state_dict(cfg, dev, synth.SEED), dev, torch.bfloat16)
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "corpus.jsonl")
synth.write_corpus_jsonl(path, bench.fast_corpus_records(bench.N_FILES, bench.N_PREMISES, synth.SEED))
corpus = Corpus(path)
N, D = len(corpus), cfg["d_model"]
E = torch.nn.functional.normalize(torch.randn(N, D, device=dev), dim=1).to(torch.bfloat16)
rng = np.random.default_rng(1)
lens = synth.synth_lengths(rng, 256, "mix", lo=16, hi=2048)
ctxs = [Context(f"M/F{int(rng.integers(2500, 5000))}.lean", f"t{j}", Pos(int(rng.integers(1, 60)), 0),
                synth.synth_state(rng, int(lens[j]) - 1)) for j in range(256)]
retr = PremiseRetriever(enc, max_seq_len=1024, num_retrieved=100)
retr.corpus, retr.corpus_embeddings, retr.embeddings_staled = corpus, E, False
tok = ByT5Tokenizer()

def predict_all():
    retr.predict_step_outputs = []
    for i in range(0, 256, 64):
        c = ctxs[i : i + 64]
        t = tok([x.serialize() for x in c], padding="longest", max_length=1024, truncation=True, return_tensors="pt")
        b = {"context": c, "context_ids": t.input_ids, "context_mask": t.attention_mask}
        for key in ("url", "commit", "file_path", "full_name", "start", "tactic_idx", "all_pos_premises"):
            b[key] = [None] * len(c)
        retr.predict_step(b, 0)

def retrieve_many():
    c = ctxs[0]
    st = synth.synth_state(np.random.default_rng(3), 100)
    for _ in range(50):
        retr.retrieve(st, c.path, c.theorem_full_name, c.theorem_pos, 100)

for name, fn in (("predict_step x4 (256 states)", predict_all), ("retrieve x50 (100-byte state)", retrieve_many)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); print(f"== {name}: {1e3 * (time.perf_counter() - t0):.1f} ms")
    pr = cProfile.Profile(); pr.enable(); fn(); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
