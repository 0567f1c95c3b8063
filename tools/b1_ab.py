"""Interleaved A/B of rp_set_option forms on the single-state retrieve() path (one process, graphs re-captured per form):
    python tools/b1_ab.py gemm_small_form 0,1,2,3 [NBYTES=100,300,1000]
Prints the median wall time per form and state size, and whether every form returned the same bits."""
import os, sys, time, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reprover_amd import _lib, synth
from reprover_amd.common import Pos
from reprover_amd.retrieval.model import PremiseRetriever
from reprover_amd.encoder import HipT5Encoder
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); _a = sys.argv; sys.argv = ["x"]; spec.loader.exec_module(bench); sys.argv = _a
opt, values = sys.argv[1].encode(), [int(v) for v in sys.argv[2].split(",")]
sizes = [int(x) for x in os.environ.get("NBYTES", "100,300,1000").split(",")]
dev = torch.device("cuda:0")
cfg = synth.t5_config("byt5-small")
model = PremiseRetriever(HipT5Encoder(cfg, bench.random_init_state_dict(cfg, dev, 1), dev), max_seq_len=2048)
d = tempfile.mkdtemp(); cp = os.path.join(d, "c.jsonl")
synth.write_corpus_jsonl(cp, bench.fast_corpus_records(5000, 130000, 1))
model.load_corpus(cp)
g = torch.Generator(device=dev); g.manual_seed(0)
model.corpus_embeddings = torch.nn.functional.normalize(torch.randn(len(model.corpus), 1472, generator=g, device=dev), dim=1).to(torch.bfloat16)
model.embeddings_staled = False
lib = _lib.load()
rng = np.random.default_rng(0)
states = {n: [synth.synth_state(rng, n) for _ in range(45)] for n in sizes}
times = {(v, n): [] for v in values for n in sizes}
answers = {}
for rnd in range(int(os.environ.get("ROUNDS", "3"))):
    for v in values:
        _lib.check(lib.rp_set_option(opt, v), "opt")
        if model._single_query is not None:
            model._single_query.clear()  # graphs hold the launches of the form they were captured under
        for n in sizes:
            for s in states[n][:5]:
                model.retrieve(s, "M/F4000.lean", "t", Pos(30, 0), 100)
            torch.cuda.synchronize()
            ts = []
            for s in states[n][5:]:
                t0 = time.perf_counter()
                prem, sc = model.retrieve(s, "M/F4000.lean", "t", Pos(30, 0), 100)
                ts.append(time.perf_counter() - t0)
            times[(v, n)].append(float(np.median(ts)))
            answers.setdefault(n, {})[v] = (tuple(p.full_name for p in prem), tuple(sc))
_lib.check(lib.rp_set_option(opt, values[0]), "opt")
for n in sizes:
    same = all(answers[n][v] == answers[n][values[0]] for v in values)
    print(f"state {n:5d} B: " + "  ".join(f"{opt.decode()}={v}: {np.median(times[(v, n)]) * 1e3:.3f} ms" for v in values) +
          f"   same bits: {same}", flush=True)
