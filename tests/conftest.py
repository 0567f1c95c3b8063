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
def tiny_weights():
    from reprover_amd import synth

    cfg = synth.t5_config("tiny")
    return cfg, synth.synth_state_dict(cfg)
