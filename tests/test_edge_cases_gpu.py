"""Edge cases of the hot path on the GPU: ragged / minimal / maximal inputs, argument errors, a
second model shape (ByT5-base geometry), the small-token-count GEMM configuration."""
import ctypes as C

import numpy as np
import pytest
import torch

import hip_helpers as hh
from oracle import common_ref, t5_ref
from reprover_amd import _lib, synth
from reprover_amd.encoder import HipT5Encoder
from reprover_amd.retrieval.model import PremiseRetriever

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny(tiny_weights):
    cfg, sd = tiny_weights
    return cfg, sd, PremiseRetriever.from_state_dict(cfg, sd, 2048, "cuda:0", dtype=torch.float32)


def _cos(a, b):
    return torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=1)


def test_ragged_minimal_and_maximal_sequences(tiny):
    cfg, sd, model = tiny
    rng = np.random.default_rng(7)
    texts = ["", "a", "ℕ", synth.synth_text(rng, 2046), synth.synth_text(rng, 2047), synth.synth_text(rng, 5000),
             synth.synth_text(rng, 127), synth.synth_text(rng, 128), synth.synth_text(rng, 255), synth.synth_text(rng, 256)]
    got = model.encode_texts(texts).cpu()
    want = t5_ref.encode_texts(cfg, sd, texts, 2048, 2)
    assert _cos(got, want).min().item() > 0.9995
    assert torch.equal(got[4], model.encode_texts([texts[4]]).cpu()[0])  # 2048 tokens incl. EOS, alone
    assert torch.allclose(got[5], model.encode_texts([texts[5][:2047]]).cpu()[0], atol=1e-6)  # truncation
    # many short sequences in one pass (more sequences than GEMM tile rows per sequence)
    many = [synth.synth_text(rng, int(n)) for n in rng.integers(1, 40, size=300)]
    got = model.encode_texts(many).cpu()
    want = t5_ref.encode_texts(cfg, sd, many, 2048, 50)
    assert _cos(got, want).min().item() > 0.9995


def test_random_batches_against_the_oracle(tiny):
    """A seeded sweep of pass shapes (1 .. 80 sequences of 1 .. 900 bytes, multi-byte symbols included): every row against
    the fp32 oracle, a random member alone against itself in the batch, the padded entry point against the packed one."""
    cfg, sd, model = tiny
    rng = np.random.default_rng(77)
    for case in range(10):
        n = int(rng.choice([1, 2, 7, 33, 80]))
        texts = [synth.synth_text(rng, int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 300, 900]))) for _ in range(n)]
        got = model.encode_texts(texts)
        want = t5_ref.encode_texts(cfg, sd, texts, 2048, 16)
        assert _cos(got.cpu(), want).min().item() > 0.9995, case
        j = int(rng.integers(0, n))
        assert (model.encode_texts([texts[j]])[0] - got[j]).abs().max().item() < 1e-6, (case, j)
        t = model.tokenizer(texts, padding="longest", max_length=2048, truncation=True, return_tensors="pt")
        assert torch.equal(model._encode(t.input_ids.cuda(), t.attention_mask.cuda()), got), case


def test_skinny_gemm_configuration_matches_default(tiny):
    cfg, sd, model = tiny
    lib = _lib.load()
    rng = np.random.default_rng(8)
    texts = [synth.synth_text(rng, 150)]  # one state: the prover's call pattern
    a = model.encode_texts(texts)
    _lib.check(lib.rp_set_option(b"gemm_skinny", 0), "opt")
    try:
        b = model.encode_texts(texts)
    finally:
        _lib.check(lib.rp_set_option(b"gemm_skinny", 1), "opt")
    assert (a - b).abs().max().item() < 1e-5  # same math, different tiling (K order unchanged)


def test_byt5_base_geometry():
    """d_model 1536, 12 heads, d_ff 3968 (ByT5-base), 3 layers to keep the CPU oracle quick."""
    cfg = synth.t5_config("byt5-base")
    cfg["num_layers"] = 3
    sd = synth.synth_state_dict(cfg, seed=5)
    model = PremiseRetriever.from_state_dict(cfg, sd, 1024, "cuda:0", dtype=torch.float32)
    rng = np.random.default_rng(9)
    texts = [synth.synth_text(rng, n) for n in (5, 77, 300, 640)]
    got = model.encode_texts(texts).cpu()
    want = t5_ref.encode_texts(cfg, sd, texts, 1024, 4)
    cos = _cos(got, want)
    print("byt5-base geometry: min cos", cos.min().item())
    assert cos.min().item() > 0.999


@pytest.mark.parametrize("B,N,D,k", [(1, 1, 32, 1), (5, 50, 64, 100), (300, 3000, 96, 1), (2, 20000, 64, 1024),
                                     (129, 17000, 160, 33)])
def test_topk_shapes(B, N, D, k):
    rng = np.random.default_rng(B + N)
    g = torch.Generator(device="cuda")
    g.manual_seed(B * N)
    E = torch.randn(N, D, generator=g, device="cuda").to(torch.bfloat16)
    Q = torch.randn(B, D, generator=g, device="cuda").to(torch.bfloat16)
    m, acc = hh.synth_masks(rng, N, B, F=max(1, min(N, 40)))
    S = (Q.double() @ E.double().T).float().cpu().numpy()
    ids, sc, cnt = hh.sim_topk(Q, E, k, hh.masks_to_device(m, Q.device))
    hh.check_topk_against_scores(ids.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy(), S, acc, k, tol=1e-4)


def test_argument_errors_are_reported_not_swallowed():
    lib = _lib.load()
    Q = torch.zeros(4, 48, dtype=torch.bfloat16, device="cuda")  # D = 48 is not a multiple of 32
    E = torch.zeros(10, 48, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(_lib.HipLibraryError, match="multiple of 32"):
        hh.sim_topk(Q, E, 5)
    Q = torch.zeros(4, 64, dtype=torch.bfloat16, device="cuda")
    E = torch.zeros(10, 64, dtype=torch.bfloat16, device="cuda")
    out = torch.empty(4, 8, device="cuda")
    st = lib.rp_sim_topk(Q.data_ptr(), E.data_ptr(), 4, 10, 64, None, None, None, 0, None, None, 0, 0, 0, out.data_ptr(),
                         out.data_ptr(), out.data_ptr(), out.data_ptr(), 1 << 20, None)
    assert st == -1 and b"k=0" in lib.rp_last_error()  # RP_E_INVALID
    with pytest.raises(_lib.HipLibraryError, match="k="):
        hh.sim_topk(Q, E, 5000)
    out = torch.empty(4, 5, device="cuda")
    st = lib.rp_sim_topk(Q.data_ptr(), E.data_ptr(), 4, 10, 64, None, None, None, 0, None, None, 0, 5, 0, out.data_ptr(),
                         out.data_ptr(), out.data_ptr(), None, 0, None)
    assert st == -3 and b"workspace" in lib.rp_last_error()  # RP_E_WORKSPACE
    cfg = synth.t5_config("tiny")
    cfg["d_kv"] = 32
    with pytest.raises(_lib.HipLibraryError, match="d_kv"):
        HipT5Encoder(cfg, synth.synth_state_dict(synth.t5_config("tiny")), "cuda:0")
