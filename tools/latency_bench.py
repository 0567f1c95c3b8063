"""B=1 retrieve() latency (the prover's call pattern, prover/tactic_generator.py:286-292)."""
import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reprover_amd import _lib, synth
from reprover_amd.common import Pos
from reprover_amd.retrieval.model import PremiseRetriever
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
bench = importlib.util.module_from_spec(spec); sys.argv = ["x"]; spec.loader.exec_module(bench)
if os.environ.get("RP_LIB_PATH"):  # another build of the library (A/B runs)
    _lib.LIB_PATH = os.path.abspath(os.environ["RP_LIB_PATH"])
dev = torch.device("cuda:0")
cfg = synth.t5_config("byt5-small")
sd = bench.random_init_state_dict(cfg, dev, 1)
from reprover_amd.encoder import HipT5Encoder
model = PremiseRetriever(HipT5Encoder(cfg, sd, dev), max_seq_len=2048)
d = tempfile.mkdtemp(); cp = os.path.join(d, "c.jsonl")
synth.write_corpus_jsonl(cp, bench.fast_corpus_records(5000, 130000, 1))
model.load_corpus(cp)
N = len(model.corpus)
g = torch.Generator(device=dev); g.manual_seed(0)
model.corpus_embeddings = torch.nn.functional.normalize(torch.randn(N, 1472, generator=g, device=dev), dim=1).to(torch.bfloat16)
model.embeddings_staled = False
rng = np.random.default_rng(0)
lib = _lib.load()
if os.environ.get('EAGER') == '1':
    model.use_graphs = False  # launch by launch: the per-class event pairs see every kernel
for _ in range(int(os.environ.get('REPEAT', '1'))):
  for nbytes in [int(x) for x in os.environ.get('NBYTES', '100,300,1000').split(',')]:
      states = [synth.synth_state(rng, nbytes) for _ in range(30)]
      for s in states[:5]:
          model.retrieve(s, "M/F4000.lean", "t", Pos(30, 0), 100)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for s in states[5:]:
          model.retrieve(s, "M/F4000.lean", "t", Pos(30, 0), 100)
      dt = (time.perf_counter() - t0) / 25  # wall time without the per-kernel event pairs
      _lib.profile_enable(True)
      for s in states[5:]:
          model.retrieve(s, "M/F4000.lean", "t", Pos(30, 0), 100)
      prof = _lib.profile_read(); _lib.profile_enable(False)
      gpu_ms = sum(v[0] for v in prof.values()) / 25
      top = sorted(prof.items(), key=lambda kv: -kv[1][0])[:9]
      print(f"state {nbytes:5d} B: retrieve() {dt*1e3:7.3f} ms wall; GPU kernels {gpu_ms:6.3f} ms; " +
            ", ".join(f"{k} {v[0]/25*1e3:.0f}us/{v[1]//25}" for k, v in top), flush=True)
if os.environ.get("CPROFILE") == "1":
    import cProfile, pstats
    states = [synth.synth_state(rng, 100) for _ in range(200)]
    for s in states[:10]:
        model.retrieve(s, "M/F4000.lean", "t", Pos(30, 0), 100)
    pr = cProfile.Profile(); pr.enable()
    for s in states[10:]:
        model.retrieve(s, "M/F4000.lean", "t", Pos(30, 0), 100)
    pr.disable()
    st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
