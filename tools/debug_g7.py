import os, sys, json, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reprover_amd import synth, _lib
from reprover_amd.common import Corpus
from reprover_amd.retrieval.model import PremiseRetriever
g = json.load(open("tests/golden/g7_predict.json"))
files = synth.synth_corpus_records(g["n_files"], g["n_premises"], seed=g["corpus_seed"], code_bytes=tuple(g["code_bytes"]))
path = os.path.join(tempfile.mkdtemp(), "c.jsonl"); synth.write_corpus_jsonl(path, files)
cfg = synth.t5_config("byt5-small"); sd = synth.synth_state_dict(cfg)
for dt in (torch.float32, torch.bfloat16):
    model = PremiseRetriever.from_state_dict(cfg, sd, 1024, "cuda:0", dtype=dt)
    corpus = Corpus(path)
    texts = [p.serialize() for p in corpus.all_premises]
    big = model.encode_texts(texts).float()
    small = torch.cat([model.encode_texts(texts[i:i+16]).float() for i in range(0, len(texts), 16)])
    cos = torch.nn.functional.cosine_similarity(big, small, dim=1)
    bad = torch.nonzero(cos < 0.999).flatten().tolist()
    ids, cu = model.tokenizer.packed(texts, 1024)
    print(dt, "rows", len(texts), "tokens", cu[-1], "bad rows", len(bad), bad[:20], "min cos", cos.min().item())
    if bad:
        print(" token offsets of bad rows:", [(int(cu[b]), int(cu[b+1])) for b in bad[:12]])
    z = np.load("tests/golden/g7_predict.npz")
    c2 = torch.nn.functional.cosine_similarity(big[:16].cpu(), torch.from_numpy(z["E_head"]), dim=1)
    c3 = torch.nn.functional.cosine_similarity(small[:16].cpu(), torch.from_numpy(z["E_head"]), dim=1)
    print("  vs golden head: big", c2.min().item(), "small", c3.min().item())
