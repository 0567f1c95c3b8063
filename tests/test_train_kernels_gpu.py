"""The training kernels in isolation, each against a plain fp32 restatement (torch autograd where a derivative is
involved), through the C ABI: wgrad GEMM (token-major operands, transposing LDS reads, split-K), the two dgrad
epilogues (gated-GELU backward, RMSNorm-backward residual update), flash attention backward incl. the bias-table
gradient.  Tolerances are one bf16 rounding of the MFMA operands (P, dS, dY are rounded to bf16 before their GEMMs)."""
import pytest
import torch

import train_helpers as th

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gen():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    g = torch.Generator(device="cuda")
    g.manual_seed(3407)
    return g


@pytest.mark.parametrize("T,ny,nx,splits", [(256, 128, 128, 1), (512, 1152, 1472, 3), (1024, 1472, 384, 4),
                                            (256, 200, 72, 2), (2048, 512, 128, 5),
                                            # negative: the 128 x 128 tile configuration
                                            (256, 128, 128, -1), (512, 1152, 1472, -2), (1024, 1472, 384, -6), (256, 200, 72, -2)])
def test_wgrad(gen, T, ny, nx, splits):
    r = th.check_wgrad(gen, T, ny, nx, splits)
    print(r)
    assert r["nan"] == 0
    # exact products of bf16 operands, fp32 accumulation over T terms
    assert r["max_err"] <= 2e-6 * T ** 0.5 * r["ref_max"] + 1e-4 and r["worst_split_err"] <= 2e-6 * T ** 0.5 * r["ref_max"] + 1e-4


@pytest.mark.parametrize("T,shapes,splits", [(1024, ((7168, 1472), (1472, 3584)), 1), (640, ((1152, 1472), (1472, 384)), 5),
                                             (256, ((64, 72), (200, 8)), 2)])
def test_wgrad_pair_launch_equals_the_single_launches(gen, T, shapes, splits):
    """The training step's weight-gradient pairs (dWi' + dWo2, dWqkv' + dWo) share one launch: per split and element the
    same K-ascending MFMA chain as the product launched alone, so the partial matrices are the same bits."""
    r = th.check_wgrad_pair(gen, T, shapes, splits)
    print(r)
    assert r["same_bits"]
    assert r["max_err"] <= 2e-6 * T ** 0.5 * r["ref_max"] + 1e-4


def test_wgrad_detects_permutations():
    assert th.check_wgrad_structured()["exact"]


@pytest.mark.parametrize("M,F,K,variant", [(256, 256, 128, 0), (512, 3584, 1472, 26), (256, 3968, 1536, 26), (256, 256, 128, 26)])
def test_geglu_backward_epilogue(gen, M, F, K, variant):
    r = th.check_geglu_bwd(gen, M, F, K, variant)
    print(r)
    assert r["nan"] == 0
    assert r["dzs_err"] <= 2 ** -8 * r["dzs_max"] + 1e-3      # bf16 store of the result
    assert r["dot_err"] <= 2e-3 * r["dot_max"] + 1e-3


def test_geglu_backward_mixed_launch_is_the_same_bits():
    """The reference's training batch gives this GEMM 14 x 41 = 574 tiles of 256 x 256 - 2.24 rounds of 256 CUs - and it
    runs as 36 token rows of full tiles + 10 rows of half tiles in one launch (plan_mixed_loose, option gemm_mixed_bwd):
    the same K-ascending chain per output element, so not a bit differs from one workgroup per full tile."""
    a = th.geglu_bwd_outputs(5, 41 * 256, 3584, 1472, 26, mixed=True)
    b = th.geglu_bwd_outputs(5, 41 * 256, 3584, 1472, 26, mixed=False)
    assert not torch.isnan(a[0].float()).any() and not torch.isnan(a[1]).any()
    assert torch.equal(a[0].view(torch.int16), b[0].view(torch.int16)) and torch.equal(a[1].view(torch.int32), b[1].view(torch.int32))


@pytest.mark.parametrize("M,N,K,variant", [(256, 128, 384, 0), (512, 1472, 7168, 26), (256, 1472, 1152, 26), (256, 128, 96, 0)])
def test_rmsnorm_backward_residual_epilogue(gen, M, N, K, variant):
    r = th.check_rms_bwd_resid(gen, M, N, K, variant)
    print(r)
    assert r["err"] <= 2 ** -15 * r["ref_max"] + 1e-4 and r["hi_err"] <= 2 ** -8 * r["ref_max"]


@pytest.mark.parametrize("lens,H", [([5, 64, 129, 300, 77], 2), ([1, 2, 3], 6), ([600, 40], 2), ([128, 256], 1)])
def test_attention_backward(gen, lens, H):
    r = th.check_attention_bwd(gen, lens, H)
    print(r)
    assert r["nan"] == 0
    assert r["att_err"] <= 2 ** -7 * r["att_max"] + 1e-3 and r["lse_err"] <= 1e-3
    for k in ("dq", "dk", "dv", "dtab"):
        assert r[k + "_err"] <= 2e-2 * r[k + "_max"] + 1e-3, (k, r)
