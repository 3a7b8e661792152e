"""Seeded synthetic inputs: frames, backward/forward flow, used by the
tests, the golden-fixture generator and bench.py.  numpy only."""
from __future__ import annotations

import numpy as np


def _blur(a: np.ndarray, sigma: float) -> np.ndarray:
    """separable Gaussian blur with reflect borders (numpy only, deterministic)."""
    r = max(1, int(3 * sigma))
    k = np.exp(-0.5 * (np.arange(-r, r + 1) / sigma) ** 2); k /= k.sum()
    for ax in (0, 1):
        pad = [(0, 0)] * a.ndim; pad[ax] = (r, r)
        p = np.pad(a, pad, mode="reflect" if a.shape[ax] > r else "edge")
        a = sum(k[i] * np.take(p, np.arange(i, i + a.shape[ax]), axis=ax) for i in range(2 * r + 1))
    return a


def smooth_frame(h: int, w: int, seed: int) -> np.ndarray:
    """uint8 RGB [H][W][3]: Gaussian-blurred noise (sigma 8 px) rescaled to 0..255."""
    rng = np.random.default_rng(seed)
    a = _blur(rng.standard_normal((h, w, 3)), 8.0)
    a = (a - a.min()) / max(a.max() - a.min(), 1e-9)
    return (a * 255.0).astype(np.uint8)


def random_frame(h: int, w: int, seed: int) -> np.ndarray:
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def backward_flow(h: int, w: int, seed: int, std: float = 2.0) -> np.ndarray:
    """[H][W][2] (u,v): blurred N(0,1) noise (sigma 16 px) scaled to `std` px + global shift U(-4,4)."""
    rng = np.random.default_rng(seed)
    f = _blur(rng.standard_normal((h, w, 2)), 16.0)
    f = f / max(f.std(), 1e-9) * std + rng.uniform(-4, 4, (1, 1, 2))
    return f.astype(np.float32)


def forward_flow_from_backward(bw: np.ndarray, seed: int, noise: float = 0.3) -> np.ndarray:
    """-backward sampled (nearest) at the displaced position + N(0, noise) px."""
    rng = np.random.default_rng(seed)
    h, w, _ = bw.shape
    ys, xs = np.mgrid[0:h, 0:w]
    sx = np.clip(np.rint(xs + bw[..., 0]).astype(int), 0, w - 1)
    sy = np.clip(np.rint(ys + bw[..., 1]).astype(int), 0, h - 1)
    fw = np.zeros_like(bw)
    # forward flow lives on the previous frame's grid: scatter -bw to where bw points
    fw[sy, sx] = -bw
    fw += rng.standard_normal(bw.shape).astype(np.float32) * noise
    return fw.astype(np.float32)


def random_flow(h: int, w: int, seed: int, scale: float = 3.0) -> np.ndarray:
    return (np.random.default_rng(seed).standard_normal((h, w, 2)) * scale).astype(np.float32)
