"""Deterministic synthetic textures for the parity tests and bench.py (no image files, no WIC).

A counter-based integer hash makes every texel a pure function of (seed, x, y, channel), so the same
image can be produced at any size on any box. The recipes follow SURVEY.md section 8d: smooth gradients
plus multi-octave value noise whose amplitude varies across the image, flat patches and hard edges, so
that 4x4 blocks range from single-colour to noisy."""
import numpy as np


def _hash32(x):
    x = np.asarray(x, np.uint64) & 0xFFFFFFFF
    x = ((x ^ (x >> 16)) * 0x7FEB352D) & 0xFFFFFFFF
    x = ((x ^ (x >> 15)) * 0x846CA68B) & 0xFFFFFFFF
    return (x ^ (x >> 16)) & 0xFFFFFFFF


def _noise(seed, xs, ys, c):
    return _hash32(xs * 73856093 + ys * 19349663 + (seed * 83492791 + c * 2654435761 + 12345))


def rgba8(width, height, seed=1, alpha="opaque"):
    """(H, W, 4) uint8. alpha: 'opaque' | 'random' | 'smooth' | 'binary'."""
    ys, xs = np.meshgrid(np.arange(height, dtype=np.uint64), np.arange(width, dtype=np.uint64), indexing="ij")
    fx, fy = xs.astype(np.float64) / max(1, width - 1), ys.astype(np.float64) / max(1, height - 1)
    out = np.zeros((height, width, 4), np.float64)
    base = [fx, fy, 0.5 * (fx + fy)]
    amp = 0.02 + 0.35 * (0.5 + 0.5 * np.sin(6.0 * fx + 2.0 * seed) * np.cos(5.0 * fy))   # flat..noisy
    for c in range(3):
        v = base[c].copy()
        for octave, scale in enumerate((1, 2, 4, 16)):
            n = _noise(seed + 101 * octave, xs // scale, ys // scale, c).astype(np.float64) / 4294967295.0
            v += amp * (n - 0.5) / (octave + 1)
        out[..., c] = v
    # flat 8x8 patches and hard vertical edges in some regions
    patch = (_noise(seed + 7, xs // 8, ys // 8, 9) & 7) == 0
    pv = _noise(seed + 11, xs // 8, ys // 8, 10)
    for c in range(3):
        out[..., c] = np.where(patch, ((pv >> (8 * c)) & 0xFF) / 255.0, out[..., c])
    edge = ((xs // 37 + ys // 53) & 3) == 0
    out[..., 0] = np.where(edge & ~patch, 1.0 - out[..., 0], out[..., 0])
    if alpha == "opaque":
        out[..., 3] = 1.0
    elif alpha == "random":
        out[..., 3] = _noise(seed + 13, xs, ys, 3).astype(np.float64) / 4294967295.0
    elif alpha == "binary":
        out[..., 3] = ((_noise(seed + 17, xs // 3, ys // 3, 3) & 3) != 0).astype(np.float64)
    else:  # smooth
        out[..., 3] = np.clip(0.5 + 0.5 * np.sin(9.0 * fx) * np.cos(7.0 * fy) + 0.1 * (_noise(seed + 19, xs, ys, 3) / 4294967295.0 - 0.5), 0, 1)
    return np.clip(np.rint(out * 255.0), 0, 255).astype(np.uint8)


def rgba16f(width, height, seed=3):
    """(H, W, 4) float16, finite, non-negative HDR data: half(exp2(uniform(-8, 6)) * smooth gradient), A = 1."""
    ys, xs = np.meshgrid(np.arange(height, dtype=np.uint64), np.arange(width, dtype=np.uint64), indexing="ij")
    fx, fy = xs.astype(np.float64) / max(1, width - 1), ys.astype(np.float64) / max(1, height - 1)
    out = np.zeros((height, width, 4), np.float64)
    e = _noise(seed, xs // 16, ys // 16, 5).astype(np.float64) / 4294967295.0 * 14.0 - 8.0
    for c in range(3):
        g = 0.25 + 0.75 * (fx if c == 0 else fy if c == 1 else 0.5 * (fx + fy))
        n = _noise(seed + 31, xs, ys, c).astype(np.float64) / 4294967295.0
        out[..., c] = np.exp2(e) * g * (0.8 + 0.4 * n)
    out[..., 3] = 1.0
    return np.minimum(out, 60000.0).astype(np.float16)
