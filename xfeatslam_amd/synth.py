"""Synthetic inputs shared by the tests, the oracle and bench.py.

Nothing here reads a dataset: TUM frames are not available (no network), so frames are
counter-based noise, box-blurred and min-max scaled to u8, and descriptor sets are
unit-norm Gaussians with a noisy permuted copy as the second set.
"""
from __future__ import annotations

import numpy as np

from .weights import uniform01


def _box3(a: np.ndarray) -> np.ndarray:
    p = np.pad(a, 1, mode="edge")
    s = np.zeros_like(a)
    for dy in range(3):
        for dx in range(3):
            s += p[dy:dy + a.shape[0], dx:dx + a.shape[1]]
    return s / 9.0


def image(H: int = 480, W: int = 640, seed: int = 42, blur: int = 2) -> np.ndarray:
    """u8 [H,W] frame: splitmix noise -> `blur` 3x3 box blurs -> min-max to 0..255."""
    a = uniform01(seed, 7, H * W).reshape(H, W)
    for _ in range(blur):
        a = _box3(a)
    lo, hi = a.min(), a.max()
    if hi <= lo:
        return np.zeros((H, W), np.uint8)
    return np.clip(np.rint((a - lo) / (hi - lo) * 255.0), 0, 255).astype(np.uint8)


IMAGE_FAMILIES = ("noise", "steps", "gradient", "lowcontrast", "saturated", "checker8", "checker4", "blobs")


def image_family(family: str, H: int = 480, W: int = 640, seed: int = 42) -> np.ndarray:
    """u8 [H,W] frames of the parity campaign: the structure box-blurred noise lacks.

    noise        `image()`
    steps        piecewise constant: 40 random axis-aligned rectangles of random grey level over a flat ground (step edges, corners,
                 T-junctions; large flat regions, where the 8x8 cells of the keypoint head see identical inputs)
    gradient     a horizontal and a vertical ramp added, quantised to u8 (every 8x8 cell differs from its neighbour by a constant)
    lowcontrast  noise squeezed into 8 grey levels around 128 (InstanceNorm amplifies the quantisation steps)
    saturated    left half of the frame 0, a band of 255, the rest noise (half the pixels identical)
    checker8     8x8-pixel squares 0 / 255 aligned with the keypoint head's cells: a frame of exactly periodic cells, the worst case
                 for ties in the 5x5 NMS and at the top-k cut
    checker4     4x4-pixel squares 32 / 224 (period 8): every cell holds the same 2x2 pattern
    blobs        smooth Gaussian bumps on a dark ground (isolated strong maxima)
    """
    if family == "noise":
        return image(H, W, seed)
    yy, xx = np.mgrid[0:H, 0:W]
    if family == "steps":
        a = np.full((H, W), 90.0)
        r = uniform01(seed, 11, 40 * 5).reshape(40, 5)
        for x0, y0, ww, hh, g in r:
            xa, ya = int(x0 * W), int(y0 * H)
            xb, yb = min(W, xa + 8 + int(ww * W / 3)), min(H, ya + 8 + int(hh * H / 3))
            a[ya:yb, xa:xb] = np.rint(g * 255.0)
        return a.astype(np.uint8)
    if family == "gradient":
        return np.clip(np.rint(xx * (200.0 / max(W - 1, 1)) + yy * (55.0 / max(H - 1, 1))), 0, 255).astype(np.uint8)
    if family == "lowcontrast":
        return (124 + (image(H, W, seed).astype(np.int32) * 8) // 256).astype(np.uint8)
    if family == "saturated":
        a = image(H, W, seed).copy()
        a[:, : W // 2] = 0
        a[:, W // 2: W // 2 + W // 8] = 255
        return a
    if family == "checker8":
        return np.where(((yy // 8) + (xx // 8)) % 2 == 0, 0, 255).astype(np.uint8)
    if family == "checker4":
        return np.where(((yy // 4) + (xx // 4)) % 2 == 0, 32, 224).astype(np.uint8)
    if family == "blobs":
        a = np.zeros((H, W))
        r = uniform01(seed, 13, 60 * 4).reshape(60, 4)
        for cx, cy, sg, amp in r:
            s2 = (2.0 + 10.0 * sg) ** 2
            a += (0.3 + amp) * np.exp(-((xx - cx * W) ** 2 + (yy - cy * H) ** 2) / (2.0 * s2))
        return np.clip(np.rint(a / a.max() * 255.0), 0, 255).astype(np.uint8)
    raise ValueError(family)


def frames(B: int, H: int = 480, W: int = 640, seed: int = 42) -> np.ndarray:
    """u8 [B,H,W]; frame i uses seed+i."""
    return np.stack([image(H, W, seed + i) for i in range(B)])


def _normal(seed: int, stream: int, n: int) -> np.ndarray:
    u1 = uniform01(seed, stream, n)
    u2 = uniform01(seed, stream + 1, n)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def descriptor_sets(n1: int = 4096, n2: int = 4096, dim: int = 64, seed: int = 0,
                    noise: float = 0.1, zero_rows: int = 0):
    """Two [n,64] f32 descriptor sets.  d1 = unit-norm Gaussian rows; d2 = a permutation
    of (d1 + noise*N(0,1)) re-normalised, truncated/extended to n2 rows, so that roughly
    half the rows are mutual nearest neighbours.  `zero_rows` trailing rows of both sets
    are zeroed, which is what `XFextractor::operator()` pads with."""
    d1 = _normal(seed, 100, n1 * dim).reshape(n1, dim)
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    m = max(n1, n2)
    base = np.resize(d1, (m, dim)) if m > n1 else d1
    d2 = base + noise * _normal(seed, 200, m * dim).reshape(m, dim)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    order = np.argsort(uniform01(seed, 300, m), kind="stable")
    d2 = d2[order][:n2]
    d1 = d1.astype(np.float32)
    d2 = d2.astype(np.float32)
    if zero_rows:
        d1[-zero_rows:] = 0
        d2[-zero_rows:] = 0
    return np.ascontiguousarray(d1), np.ascontiguousarray(d2)
