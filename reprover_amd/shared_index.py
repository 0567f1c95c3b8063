"""One device-resident index per GPU, shared by every worker process on that GPU.

The reference gives each prover worker its own tactic generator, hence its own retriever and its own copy of the
corpus embeddings on the GPU (prover/proof_search.py:438-447 creates one ``GpuProver`` Ray actor per worker; each
loads the indexed corpus, retrieval/model.py:81-85, and keeps ``corpus_embeddings`` on its device, :363-366).  At
130 000 x 1472 bf16 that is 383 MB per worker plus the per-premise mask arrays; the search itself only READS them.

Here one process (the owner) holds the matrix and the mask arrays, ``export_index`` turns them into a picklable
handle (HIP IPC memory handles, through torch's CUDA-tensor sharing: device memory is what torch is here for), and
every worker maps the same HBM with ``attach_index``.  Nothing is copied and the C ABI is untouched: it takes plain
device pointers and does not care which process allocated them.

Rules of HIP IPC that the caller inherits:
  * the owner must outlive the workers' use of the mapping and must keep the exported tensors alive;
  * the mapping is read-only by convention (``reindex_corpus`` in a worker would write into everybody's index; an
    attached retriever refuses to);
  * ``HSA_ENABLE_IPC_MODE_LEGACY=0`` must be set in every process (this image exports it).
"""
import pickle
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from .common import Corpus, Fp8Index


@dataclass
class SharedIndexHandle:
    """Picklable description of an index resident in another process's HBM."""

    payload: bytes  # pickled {name: (rebuild_fn, args)} of torch.multiprocessing's CUDA reductions
    n_premises: int
    d_model: int
    device_index: int
    has_fp8: bool


def _reduce(t: torch.Tensor):
    from torch.multiprocessing.reductions import reduce_tensor

    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("only contiguous device tensors can be shared")
    return reduce_tensor(t)


def export_index(embeddings: torch.Tensor, corpus: Corpus, fp8: Optional[Fp8Index] = None) -> SharedIndexHandle:
    """Owner side.  ``embeddings``: the bf16 [N, D] matrix on the GPU (what ``rp_sim_topk`` streams)."""
    if embeddings.dtype != torch.bfloat16 or not embeddings.is_cuda:
        raise ValueError("export the device-resident bf16 matrix (PremiseRetriever.share_index prepares it)")
    file_of, end_key = corpus._device_arrays(embeddings.device)
    tensors: Dict[str, torch.Tensor] = {"embeddings": embeddings, "file_of": file_of, "end_key": end_key}
    if fp8 is not None:
        tensors["fp8_codes"], tensors["fp8_scale"] = fp8.codes, fp8.scale
    payload = pickle.dumps({k: _reduce(v) for k, v in tensors.items()})
    return SharedIndexHandle(payload, int(embeddings.shape[0]), int(embeddings.shape[1]),
                             embeddings.device.index or 0, fp8 is not None)


def attach_index(handle: SharedIndexHandle) -> Dict[str, torch.Tensor]:
    """Worker side: tensors over the owner's HBM (same GPU).  Keep them referenced while in use."""
    torch.cuda.init()  # the rebuild functions expect an initialised device runtime
    out = {}
    for name, (fn, args) in pickle.loads(handle.payload).items():
        out[name] = fn(*args)
    e = out["embeddings"]
    assert tuple(e.shape) == (handle.n_premises, handle.d_model) and e.dtype == torch.bfloat16
    return out
