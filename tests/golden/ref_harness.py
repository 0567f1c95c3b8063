"""Authoring-container-only harness that imports the *reference* (lean-dojo/ReProver at
/root/reference) plus HuggingFace `transformers`, so golden vectors can be generated from the
reference's own code.  Never imported by product code, tests, smoke() or bench.py; never run on
the GPU box (there is no /root/reference there).

The reference needs `lean_dojo`, `loguru`, `pytorch_lightning`, `deepspeed`, none of which exist
in this image.  Only the attributes the retrieval path touches are stood in for (SURVEY.md
App. C); none of these stand-ins implements any arithmetic of the path.
"""
import dataclasses
import functools
import sys
import types

import torch
import transformers  # noqa: F401  (resolve lazy modules before stubbing deepspeed)
from transformers import get_constant_schedule_with_warmup  # noqa: F401
from transformers import AutoModelForTextEncoding, AutoTokenizer  # noqa: F401

REFERENCE_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


@functools.total_ordering
@dataclasses.dataclass(frozen=True)
class Pos:
    """Stand-in for lean_dojo.Pos: (line_nb, column_nb), iterable, lexicographic order."""

    line_nb: int
    column_nb: int

    def __iter__(self):
        yield self.line_nb
        yield self.column_nb

    def __lt__(self, o):
        return (self.line_nb, self.column_nb) < (o.line_nb, o.column_nb)


class _Quiet:
    def __getattr__(self, k):
        return lambda *a, **kw: None


class _LightningModule(torch.nn.Module):
    def save_hyperparameters(self):
        pass

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def trainer(self):
        raise RuntimeError("not attached to a trainer")


def install():
    """Install the stand-in modules and import the reference. Returns (common, retrieval.model)."""
    _mod("lean_dojo", Pos=Pos, LeanGitRepo=object)
    _mod("lean_dojo.data_extraction")
    _mod("lean_dojo.data_extraction.lean", Pos=Pos)  # the module lean_dojo defines Pos in (what a pickle of it names)
    _mod("loguru", logger=_Quiet())
    _mod("pytorch_lightning", LightningModule=_LightningModule, LightningDataModule=object, Trainer=object)
    _mod("pytorch_lightning.utilities")
    _mod("pytorch_lightning.utilities.deepspeed", convert_zero_checkpoint_to_fp32_state_dict=None)
    _mod("pytorch_lightning.strategies")
    _mod("pytorch_lightning.strategies.deepspeed", DeepSpeedStrategy=type("DeepSpeedStrategy", (), {}))
    _mod("deepspeed")
    _mod("deepspeed.ops")
    _mod("deepspeed.ops.adam", FusedAdam=None, DeepSpeedCPUAdam=None)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import common  # noqa: E402
    import retrieval.model as rm  # noqa: E402  (sets matmul precision "medium")

    torch.set_float32_matmul_precision("highest")
    return common, rm


def offline_retriever(rm, t5_config_kwargs, state_dict=None, max_seq_len=2048):
    """Build the reference PremiseRetriever around a config-constructed T5EncoderModel
    (no network): patch the two from_pretrained calls at retrieval/model.py:44-45."""
    from transformers import ByT5Tokenizer, T5Config, T5EncoderModel

    cfg = T5Config(**t5_config_kwargs)

    def _tok(_name):
        return ByT5Tokenizer()

    def _enc(_name):
        m = T5EncoderModel(cfg)
        if state_dict is not None:
            missing, unexpected = m.load_state_dict(state_dict, strict=False)
            assert not unexpected, unexpected
        return m

    rm.AutoTokenizer.from_pretrained = staticmethod(_tok)
    rm.AutoModelForTextEncoding.from_pretrained = staticmethod(_enc)
    model = rm.PremiseRetriever.load_hf("offline", max_seq_len, torch.device("cpu"))
    return model
