"""CPU oracle for the e4m3 index (BASELINE.json configs[4]) — TEST INFRASTRUCTURE ONLY.

The reference has no fp8 path (it keeps the index in the model dtype, retrieval/model.py:190-194);
this restates the definition `include/reprover_hip.h` gives for `rp_quantize_rows_e4m3` /
`rp_sim_topk_fp8` in plain numpy, and is pinned against PyTorch's own `float8_e4m3fn` cast
(tests/test_fp8_cpu.py).  OCP e4m3fn: sign, 4 exponent bits (bias 7), 3 mantissa bits; largest
finite 448 (0x7E), 0x7F = NaN, smallest normal 2^-6, subnormals k * 2^-9.
"""
from __future__ import annotations

import numpy as np

E4M3_MAX = np.float32(448.0)


def encode_e4m3(y: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> e4m3fn codes (uint8), saturating at +-448."""
    y = np.asarray(y, dtype=np.float32)
    sign = ((y.view(np.uint32) >> np.uint32(24)) & np.uint32(0x80)).astype(np.uint32)
    a = np.minimum(np.abs(y), E4M3_MAX).astype(np.float32)
    a = np.where(np.isnan(a), E4M3_MAX, a).astype(np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    u = u + np.uint64(0x7FFFF) + ((u >> np.uint64(20)) & np.uint64(1))
    normal = ((((u >> np.uint64(23)) - np.uint64(120)) << np.uint64(3)) | ((u >> np.uint64(20)) & np.uint64(7)))
    normal = np.minimum(normal, np.uint64(0x7E))
    sub = np.rint(a * np.float32(512.0)).astype(np.uint64)  # numpy rint = round half to even
    code = np.where(a >= np.float32(0.015625), normal, sub).astype(np.uint32)
    return (sign | code).astype(np.uint8)


def decode_e4m3(codes: np.ndarray) -> np.ndarray:
    """e4m3fn codes -> fp32 (exact).  0x7F / 0xFF decode to NaN."""
    c = np.asarray(codes, dtype=np.uint8).astype(np.int32)
    sign = np.where(c & 0x80, -1.0, 1.0)
    e = (c >> 3) & 0xF
    m = c & 7
    val = np.where(e == 0, m * 2.0 ** -9, (1.0 + m / 8.0) * np.exp2(e.astype(np.float64) - 7.0))
    val = np.where((c & 0x7F) == 0x7F, np.nan, val)
    return (sign * val).astype(np.float32)


def quantize_rows_e4m3(X: np.ndarray):
    """(codes uint8 [R, D], scale f32 [R]) as rp_quantize_rows_e4m3 defines them."""
    X = np.asarray(X, dtype=np.float32)
    amax = np.abs(X).max(axis=1).astype(np.float32)
    nz = amax > 0
    inv = np.where(nz, E4M3_MAX / np.where(nz, amax, np.float32(1)), np.float32(0)).astype(np.float32)
    scale = np.where(nz, amax / E4M3_MAX, np.float32(1)).astype(np.float32)
    y = (X * inv[:, None]).astype(np.float32)
    return encode_e4m3(y), scale


def scores_fp8(Q8: np.ndarray, q_scale: np.ndarray, E8: np.ndarray, e_scale: np.ndarray) -> np.ndarray:
    """fp32 [B, N]: (sum_c q8 * e8, exact) * q_scale * e_scale, multiplications rounded to fp32 in that
    order.  The products of two e4m3 values are exact in fp32 and their fp64 sum is exact, so this is the
    infinitely precise value of what the MFMA accumulates in fp32."""
    acc = decode_e4m3(Q8).astype(np.float64) @ decode_e4m3(E8).astype(np.float64).T
    s = acc.astype(np.float32) * np.asarray(q_scale, np.float32)[:, None]
    return (s * np.asarray(e_scale, np.float32)[None, :]).astype(np.float32)
