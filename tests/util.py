"""Helpers shared by the tests."""
import numpy as np

NP_REAL = {"f64": np.float64, "f32": np.float32}


def random_pair(H, W, seed=0):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 256, size=(H, W), dtype=np.uint8),
            rng.integers(0, 256, size=(H, W), dtype=np.uint8))


def valid_mask(H, W, D):
    """[H][W][4][D] bool: slot exists and d <= j."""
    m = np.zeros((H, W, 4, D), dtype=bool)
    jj = np.arange(W)[None, :, None]
    dd = np.arange(D)[None, None, :]
    ok = np.broadcast_to(dd <= jj, (H, W, D))
    ii = np.arange(H)[:, None, None]
    m[:, :, 0, :] = ok & (jj > 0)
    m[:, :, 1, :] = ok & (ii > 0)
    m[:, :, 2, :] = ok & (jj + 1 < W)
    m[:, :, 3, :] = ok & (ii + 1 < H)
    return m


def random_phi(H, W, D, dtype, seed=0, scale=10.0):
    """Random potentials on existing slots / valid labels, 0 elsewhere (the
    reference never writes the other entries)."""
    rng = np.random.default_rng(seed)
    phi = (rng.standard_normal((H, W, 4, D)) * scale).astype(NP_REAL[dtype])
    phi[~valid_mask(H, W, D)] = 0
    return phi


def rel_err(a, b):
    """max |a-b| / max(1, max|b|): norm-wise relative error."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max()))
