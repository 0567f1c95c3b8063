"""ctypes binding of libreprover_hip.so (include/reprover_hip.h).

This is the reference-side FFI stub: tensors are passed as raw device
pointers (``tensor.data_ptr()``) plus sizes; the current torch HIP stream is passed as
``void*``.  There is no CPU fallback: if the library is missing or a call fails, an exception
is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libreprover_hip.so")

RP_OK = 0
RP_DT_F32, RP_DT_BF16 = 0, 1
RP_TOPK_AUTO, RP_TOPK_DENSE = 0, 1
RP_EPI_STORE_BF16, RP_EPI_RESID, RP_EPI_GEGLU_BF16, RP_EPI_RESID8 = 0, 1, 2, 3
ABI_VERSION = 6  # round 5 added exports, enum values and changed the workspace planes without a bump: 5 was skipped
KERNEL_CLASSES = ["embed", "rmsnorm", "gemm_qkv", "attention", "gemm_o", "gemm_wi", "gemm_wo", "pool", "scan",
                  "select", "scan_sample", "bwd_dgrad", "bwd_wgrad", "bwd_attention", "bwd_other", "optimizer", "collective"]


class RpT5Config(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int32),
        ("d_model", C.c_int32),
        ("d_kv", C.c_int32),
        ("num_heads", C.c_int32),
        ("d_ff", C.c_int32),
        ("num_layers", C.c_int32),
        ("rel_num_buckets", C.c_int32),
        ("rel_max_distance", C.c_int32),
        ("layer_norm_eps", C.c_float),
    ]


class RpT5LayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln_attn", "q", "k", "v", "o", "ln_ff", "wi_0", "wi_1", "wo")]


class RpT5Weights(C.Structure):
    _fields_ = [
        ("embed", C.c_void_p),
        ("rel_bias", C.c_void_p),
        ("final_ln", C.c_void_p),
        ("layers", C.POINTER(RpT5LayerWeights)),
    ]


class HipLibraryError(RuntimeError):
    """A libreprover_hip call returned a non-zero status."""


# name -> (restype, argtypes); every symbol declared in include/reprover_hip.h
SIGNATURES = {
    "rp_abi_version": (C.c_int32, []),
    "rp_last_error": (C.c_char_p, []),
    "rp_encoder_create": (C.c_int32, [C.POINTER(RpT5Config), C.POINTER(RpT5Weights), C.c_int32, C.POINTER(C.c_void_p)]),
    "rp_encoder_destroy": (None, [C.c_void_p]),
    "rp_encoder_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32]),
    "rp_encode_varlen": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
         C.c_size_t, C.c_void_p],
    ),
    "rp_encode_padded_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32]),
    "rp_encode_padded": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
         C.c_size_t, C.c_void_p],
    ),
    "rp_relative_position_bucket": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "rp_sim_topk_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "rp_sim_topk": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
         C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
         C.c_size_t, C.c_void_p],
    ),
    "rp_sim_topk_after": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
         C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
         C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "rp_quantize_rows_e4m3": (
        C.c_int32,
        [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "rp_sim_topk_fp8": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
         C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
         C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "rp_sim_topk_fp8_after": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
         C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "rp_topk_merge_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "rp_topk_merge": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
         C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "rp_topk_merge_strided": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
         C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "rp_comm_unique_id": (C.c_int32, [C.c_void_p]),
    "rp_comm_init": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "rp_comm_destroy": (C.c_int32, [C.c_void_p]),
    "rp_comm_world": (C.c_int32, [C.c_void_p]),
    "rp_comm_rank": (C.c_int32, [C.c_void_p]),
    "rp_comm_allgather": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rp_allgather_topk": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
         C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "rp_build_file_bits": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "rp_contrastive_mse_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "rp_contrastive_mse": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
         C.c_size_t, C.c_void_p],
    ),
    "rp_contrastive_mse_backward": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
         C.c_void_p],
    ),
    "rp_adamw_step": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
         C.c_float, C.c_void_p],
    ),
    "rp_adamw_step_clipped": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
         C.c_float, C.c_void_p, C.c_float, C.c_void_p],
    ),
    "rp_train_param_tensors": (C.c_int32, [C.POINTER(RpT5Config)]),
    "rp_train_param_layout": (C.c_int32, [C.POINTER(RpT5Config), C.POINTER(C.c_int64)]),
    "rp_trainer_create": (C.c_int32, [C.POINTER(RpT5Config), C.c_void_p, C.POINTER(C.c_void_p)]),
    "rp_trainer_destroy": (None, [C.c_void_p]),
    "rp_trainer_encoder": (C.c_void_p, [C.c_void_p]),
    "rp_trainer_load_params": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rp_trainer_set_dropout": (C.c_int32, [C.c_void_p, C.c_float, C.c_uint32]),
    "rp_dbg_dropout_mask": (
        C.c_int32,
        [C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p],
    ),
    "rp_train_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32]),
    "rp_train_forward": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "rp_train_backward": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
         C.c_size_t, C.c_void_p],
    ),
    "rp_grad_norm": (C.c_int32, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rp_dbg_wgrad": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "rp_dbg_wgrad_pair": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "rp_dbg_attention_bwd": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "rp_dbg_dgrad": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
         C.c_void_p, C.c_int32, C.c_void_p],
    ),
    "rp_dbg_gemm": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    ),
    "rp_dbg_gemm_fused": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
         C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p],
    ),
    "rp_dbg_mfma_probe": (C.c_int32, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "rp_dbg_rowscale": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p]),
    "rp_dbg_attention": (
        C.c_int32,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    ),
    "rp_set_option": (C.c_int32, [C.c_char_p, C.c_int32]),
    "rp_profile_enable": (C.c_int32, [C.c_int32]),
    "rp_profile_read": (C.c_int32, [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the engine (once).  Raises if it has not been built — never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    # The engine shares torch's HIP runtime instance (device pointers and streams cross the ABI),
    # so torch's bundled libamdhip64 must be the one already loaded when the library is dlopen'ed.
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m reprover_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the retrieval path."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    got = lib.rp_abi_version()
    if got != ABI_VERSION:
        raise HipLibraryError(f"libreprover_hip ABI {got} != binding {ABI_VERSION}; rebuild the library")
    _lib = lib
    # experiments: RP_OPTIONS="name=value,name=value" -> rp_set_option (tuning knobs only; unknown names raise)
    for item in filter(None, os.environ.get("RP_OPTIONS", "").split(",")):
        name, _, value = item.partition("=")
        check(lib.rp_set_option(name.strip().encode(), int(value)), f"RP_OPTIONS {item}")
    return lib


def check(status: int, what: str) -> None:
    if status != RP_OK:
        msg = load().rp_last_error().decode(errors="replace")
        raise HipLibraryError(f"{what} failed with status {status}: {msg}")


def ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (None passes NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "libreprover_hip takes contiguous device tensors"
    return t.data_ptr()


def current_stream() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream


def profile_enable(on: bool) -> None:
    check(load().rp_profile_enable(1 if on else 0), "rp_profile_enable")


def profile_read() -> dict:
    """{kernel class: (total_ms, launches)} of the launches recorded since profile_enable(True)."""
    lib = load()
    out = {}
    for i, name in enumerate(KERNEL_CLASSES):
        ms, n = C.c_double(), C.c_int64()
        check(lib.rp_profile_read(i, C.byref(ms), C.byref(n)), "rp_profile_read")
        out[name] = (ms.value, n.value)
    return out
