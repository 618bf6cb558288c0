"""Deterministic synthetic stereo pairs for the benchmark and the large-size tests.

The reference ships only 5x5 toy images and its `main` reads a 400x400 pair that
is not bundled (diffusion_stereo/src/main.rs:37-40), so benchmark inputs are
generated: a smooth random right image, a piecewise-planar ground-truth disparity
that respects the reference's label constraints (d <= j and d[j+1] <= d[j] + 1,
diffusion_stereo/src/diffusion_graph.rs:152-169, utils.rs:121-130), and a left
image warped from the right one plus small noise.
"""
import numpy as np


def _hash32(x, seed):
    x = (x.astype(np.uint32) * np.uint32(2654435761)) ^ np.uint32(seed)
    for _ in range(3):
        x ^= x << np.uint32(13)
        x ^= x >> np.uint32(17)
        x ^= x << np.uint32(5)
    return x


def synthetic_disparity(H, W, D):
    """[H][W] int32 ground truth in [0, D-1], d <= j, d[j+1] <= d[j] + 1."""
    ii, jj = np.mgrid[0:H, 0:W]
    bh, bw = max(H // 3, 1), max(W // 4, 1)
    region = (ii // bh) * 7 + (jj // bw)
    base = (_hash32(region, 0x5EED0003) % np.uint32(max(D, 1))).astype(np.float64) * 0.7
    slope = ((_hash32(region, 0x5EED0004) % np.uint32(64)).astype(np.float64) - 32.0) / 512.0
    d = np.rint(base + slope * (jj % bw)).astype(np.int64)
    d = np.clip(d, 0, D - 1)
    # ordering constraint: d[j] <= d[k] + (j - k) for all k <= j, and d[j] <= j (k = "column -1" with d = -1... use j)
    d = np.minimum(d, jj)
    d = jj + np.minimum.accumulate(d - jj, axis=1)
    return d.astype(np.int32)


def synthetic_pair(H, W, D, seed=0x5EED0001):
    """Returns (left u8 [H][W], right u8 [H][W], ground-truth disparity int32 [H][W])."""
    idx = np.arange(H * W, dtype=np.uint32).reshape(H, W)
    tex = (_hash32(idx, seed) & np.uint32(0xFF)).astype(np.float32)
    pad = np.pad(tex, 1, mode="edge")
    blur = sum(pad[a:a + H, b:b + W] for a in range(3) for b in range(3)) / 9.0
    # stretch contrast back after the blur so that costs use the 0..255 range
    right = np.clip((blur - 127.5) * 2.5 + 127.5, 0, 255).astype(np.uint8)
    d = synthetic_disparity(H, W, D)
    jj = np.arange(W, dtype=np.int64)[None, :]
    src = jj - d
    left = np.take_along_axis(right, src, axis=1).astype(np.int32)
    noise = (_hash32(idx, seed + 1) % np.uint32(5)).astype(np.int32) - 2
    left = np.clip(left + noise, 0, 255).astype(np.uint8)
    return left, right, d
