import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_PARITY_MARGINS = {}


@pytest.fixture(scope="session")
def parity_margins():
    """Filled by the end-to-end GPU parity tests (G5 / G7 / G9 and G5h / G7h / G9h); written to profiles/r06_parity_margins.json and
    gpurun_out/parity_margins.json and printed in the terminal summary when the session ends."""
    return _PARITY_MARGINS


def pytest_sessionfinish(session, exitstatus):
    if _PARITY_MARGINS:
        from oracle.parity_margins import write_margins

        write_margins(_PARITY_MARGINS, ROOT)


def pytest_terminal_summary(terminalreporter):
    if not _PARITY_MARGINS:
        return
    w = terminalreporter.write_line
    w("parity margins vs the reference's fp32 goldens (HF-bf16 on the same inputs in brackets):")
    for name, m in sorted(_PARITY_MARGINS.items()):
        if "min_row_cosine" in m and "max_abs_pairwise_score_err" in m:
            w(f"  {name}: min row cosine {m['min_row_cosine']:.5f} [{m['hf_bf16_min_row_cosine']:.5f}], max |d emb| "
              f"{m['max_abs_emb_err']:.2e} [{m['hf_bf16_max_abs_emb_err']:.2e}], max |d score| {m['max_abs_pairwise_score_err']:.2e} "
              f"[{m['hf_bf16_max_abs_pairwise_score_err']:.2e}] rms {m['rms_pairwise_score_err']:.2e} "
              f"[{m['hf_bf16_rms_pairwise_score_err']:.2e}], written contract met: {m['contract_met']}, no worse than "
              f"HF-bf16 on any metric: {m['no_worse_than_hf_bf16']}")
        elif "min_row_cosine" in m:
            w(f"  {name}: {m['rows']} rows, min cosine {m['min_row_cosine']:.5f} [{m['hf_bf16_min_row_cosine']:.5f}], mean "
              f"{m['mean_row_cosine']:.6f}, max |d emb| {m['max_abs_emb_err']:.2e}")
        else:
            k = m["k"]
            w(f"  {name}: max |d score| {m['max_abs_score_err']:.2e} [{m['hf_bf16_max_abs_score_err']:.2e}], gap-rule ranks "
              f"{m['gap_rule_ranks_checked']} of {m['ranks']} checked / {m['gap_rule_mismatches']} mismatched, top-1 "
              f"{m['top1_agreement']:.3f}" + (f" [{m['hf_bf16_top1_agreement']:.3f}]" if "hf_bf16_top1_agreement" in m else "") +
              f", top-{k} overlap {m[f'top{k}_overlap']:.3f}" +
              (f" [{m[f'hf_bf16_top{k}_overlap']:.3f}]" if f"hf_bf16_top{k}_overlap" in m else "") +
              f", written contract met: {m['contract_met']}")


@pytest.fixture(scope="session")
def hip_lib():
    """The C-ABI engine, built in-tree (cross-compiles without a GPU)."""
    from reprover_amd import _lib, build

    build.build()
    return _lib.load()


@pytest.fixture(scope="session")
def small_weights():
    """(cfg, HF-keyed fp32 state dict) of the synthetic ByT5-small used by fixtures G5/G7."""
    from reprover_amd import synth

    cfg = synth.t5_config("byt5-small")
    return cfg, synth.synth_state_dict(cfg)


@pytest.fixture(scope="session")
def small_weights_hf():
    """ByT5-small weights at exactly HF's init scales (SURVEY.md 8c's G5 recipe): fixtures G5h / G7h, on which the written
    floating-point contract is asserted un-relaxed."""
    from reprover_amd import synth

    cfg = synth.t5_config("byt5-small")
    return cfg, synth.synth_state_dict(cfg, scale="hf")


@pytest.fixture(scope="session")
def tiny_weights():
    from reprover_amd import synth

    cfg = synth.t5_config("tiny")
    return cfg, synth.synth_state_dict(cfg)
