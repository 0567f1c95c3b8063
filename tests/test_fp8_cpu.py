"""The e4m3 oracle (oracle/fp8_ref.py) against PyTorch's own float8_e4m3fn cast; CPU only."""
import numpy as np
import torch

from oracle import fp8_ref


def _torch_encode(y: np.ndarray) -> np.ndarray:
    return torch.from_numpy(y).to(torch.float8_e4m3fn).view(torch.uint8).numpy()


def test_every_code_round_trips():
    codes = np.arange(256, dtype=np.uint8)
    vals = fp8_ref.decode_e4m3(codes)
    finite = ~np.isnan(vals)
    assert finite.sum() == 254 and vals[0x7E] == 448.0 and vals[0x08] == 2.0 ** -6 and vals[0x01] == 2.0 ** -9
    back = fp8_ref.encode_e4m3(vals[finite])
    want = codes[finite].copy()
    assert np.array_equal(back & 0x7F, want & 0x7F)
    assert np.array_equal(back[want != 0x80], want[want != 0x80])  # -0 keeps its sign bit too
    assert np.array_equal(torch.from_numpy(codes[finite]).view(torch.float8_e4m3fn).float().numpy(), vals[finite])


def test_encode_matches_torch_cast_incl_midpoints_and_subnormals():
    rng = np.random.default_rng(0)
    vals = fp8_ref.decode_e4m3(np.arange(0, 0x7F, dtype=np.uint8))  # 0 .. 448 ascending
    mids = ((vals[:-1].astype(np.float64) + vals[1:].astype(np.float64)) / 2).astype(np.float32)  # exact ties
    near = np.concatenate([np.nextafter(mids, np.float32(0)), np.nextafter(mids, np.float32(1e9))])
    y = np.concatenate([vals, mids, near, rng.uniform(-448, 448, 20000).astype(np.float32),
                        (rng.standard_normal(20000) * 0.02).astype(np.float32),
                        (rng.standard_normal(5000) * 1e-3).astype(np.float32)]).astype(np.float32)
    y = np.concatenate([y, -y])
    assert np.array_equal(fp8_ref.encode_e4m3(y), _torch_encode(y))


def test_saturation_is_ours_not_torchs():
    # values above 448 never occur after row scaling; the definition saturates them
    assert fp8_ref.encode_e4m3(np.array([1e6, -1e6, 464.0, 480.0], np.float32)).tolist() == [0x7E, 0xFE, 0x7E, 0x7E]


def test_quantize_rows_definition():
    rng = np.random.default_rng(1)
    X = rng.standard_normal((64, 192)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    X[5] = 0.0
    X[6, :] = 0.0
    X[6, 17] = -3.0
    codes, scale = fp8_ref.quantize_rows_e4m3(X)
    assert scale[5] == 1.0 and not codes[5].any()
    assert codes[6, 17] == 0xFE and scale[6] == np.float32(3.0) / np.float32(448.0)
    deq = fp8_ref.decode_e4m3(codes) * scale[:, None]
    # 3 mantissa bits: relative error <= 2^-4 for normals; tiny entries fall into the subnormal grid
    err = np.abs(deq - X)
    assert (err <= np.maximum(np.abs(X) * 2.0 ** -4, scale[:, None] * 2.0 ** -10) + 1e-12).all()
    S = fp8_ref.scores_fp8(codes[:8], scale[:8], codes, scale)
    assert np.abs(S - X[:8] @ X.T).max() < 2e-2
