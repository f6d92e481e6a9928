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


# ---- SURVEY.md section 8d recipes: LCG-driven images of BASELINE.json's configurations ------------------------------------
# s <- s * 1664525 + 1013904223 (mod 2^32), byte = s >> 24. The n-th state is computed by jump-ahead (doubling), so a plane of
# any size costs log2(n) numpy passes and the same (seed, index) always gives the same byte.
_LCG_A, _LCG_C = 1664525, 1013904223


def lcg_bytes(seed, n):
    """First n bytes (s_1 >> 24, s_2 >> 24, ...) of the LCG started at s_0 = seed."""
    n = int(n)
    s = np.empty(n + 1, np.uint64)
    s[0] = seed & 0xFFFFFFFF
    have, a, c = 1, _LCG_A, _LCG_C            # s[i + have] = a * s[i] + c
    while have < n + 1:
        take = min(have, n + 1 - have)
        s[have:have + take] = (s[:take] * np.uint64(a) + np.uint64(c)) & np.uint64(0xFFFFFFFF)
        c = (a * c + c) & 0xFFFFFFFF
        a = (a * a) & 0xFFFFFFFF
        have += take
    return (s[1:] >> np.uint64(24)).astype(np.uint8)


def _lcg_plane(seed, height, width, cell):
    """(height, width) bytes: one LCG byte per cell x cell square, replicated."""
    ch, cw = (height + cell - 1) // cell, (width + cell - 1) // cell
    p = lcg_bytes(seed, ch * cw).reshape(ch, cw)
    if cell > 1:
        p = np.repeat(np.repeat(p, cell, axis=0), cell, axis=1)
    return p[:height, :width]


def survey_rgba8(width, height, seed=2, alpha="opaque"):
    """SURVEY.md section 8d, cfg1 / cfg2 / cfg4 recipe: R = x-gradient + 4-bit noise, G = y-gradient + 4-bit noise,
    B = diagonal + 5-bit noise, the noise summed over 4 octaves (cells of 1, 2, 4, 8 texels) and scaled per 64 x 64 region
    by a gain of 0, 1, 2 or 4 drawn from the same LCG, so that 4 x 4 blocks range from flat (pure gradient) to noisy.
    alpha: 'opaque' (255) | 'random' (one LCG byte per texel). (H, W, 4) uint8."""
    xs = np.arange(width, dtype=np.int64)[None, :]
    ys = np.arange(height, dtype=np.int64)[:, None]
    base = [(xs * 255 // max(1, width - 1)).astype(np.int16), (ys * 255 // max(1, height - 1)).astype(np.int16),
            ((xs + ys) * 255 // max(1, width + height - 2)).astype(np.int16)]
    gain = np.array([0, 1, 2, 4], np.int16)[_lcg_plane(seed * 7919 + 5, height, width, 64) >> 6]
    out = np.empty((height, width, 4), np.uint8)
    for c in range(3):
        bits = 5 if c == 2 else 4
        acc = np.zeros((height, width), np.int16)
        for octave in range(4):
            acc += (_lcg_plane(seed * 7919 + 101 * (octave * 3 + c) + 11, height, width, 1 << octave) >> (8 - bits)).astype(np.int16)
        acc -= 4 << (bits - 1)
        acc *= gain
        acc >>= 1
        acc += base[c]
        out[..., c] = np.clip(acc, 0, 255)
    out[..., 3] = 255 if alpha == "opaque" else _lcg_plane(seed * 7919 + 977, height, width, 1)
    return out


def survey_rgba16f(width, height, seed=3):
    """SURVEY.md section 8d, cfg3 recipe: half(exp2(uniform(-8, 6)) * smooth gradient), finite, non-negative, A = 1. The exponent
    is drawn per 8 x 8 cell (one LCG byte -> [-8, 6]), every texel gets a factor in [0.75, 1.25) from its own LCG byte."""
    ys, xs = np.meshgrid(np.arange(height, dtype=np.float64), np.arange(width, dtype=np.float64), indexing="ij")
    fx, fy = xs / max(1, width - 1), ys / max(1, height - 1)
    e = _lcg_plane(seed * 7919 + 3, height, width, 8).astype(np.float64) * (14.0 / 255.0) - 8.0
    out = np.empty((height, width, 4), np.float64)
    for c in range(3):
        g = 0.25 + 0.75 * (fx if c == 0 else fy if c == 1 else 0.5 * (fx + fy))
        n = _lcg_plane(seed * 7919 + 31 * (c + 1), height, width, 1).astype(np.float64) / 256.0
        out[..., c] = np.exp2(e) * g * (0.75 + 0.5 * n)
    out[..., 3] = 1.0
    return np.minimum(out, 60000.0).astype(np.float16)


def huge_rgba8(width, height, seed, noisy_block_rows, alpha="opaque"):
    """An image of the sizes DirectXTexImage.cpp:127-131 allows (up to 16384^2) that the reference still encodes in minutes: every
    4 x 4 block is one flat colour (a hash of its position: the encoders' early-outs), except the block rows listed in
    `noisy_block_rows` (iterable of block-row indices), which carry the SURVEY 8d recipe (survey_rgba8 of that strip, seed + block row).
    Put the noisy rows where the path has a seam: the first and last rows, around a multiple of 65536 texel rows' worth of bytes,
    around block 2^22 (the BC6H / BC7 pass boundary). (H, W, 4) uint8."""
    nbw, nbh = (width + 3) // 4, (height + 3) // 4
    by, bx = np.meshgrid(np.arange(nbh, dtype=np.uint64), np.arange(nbw, dtype=np.uint64), indexing="ij")
    h = _hash32(bx * 2654435761 + by * 40503 + seed * 97 + 1)
    flat = np.empty((nbh, nbw, 4), np.uint8)
    flat[..., 0] = h & 0xFF; flat[..., 1] = (h >> 8) & 0xFF; flat[..., 2] = (h >> 16) & 0xFF
    flat[..., 3] = 255 if alpha == "opaque" else (h >> 24) & 0xFF
    out = np.repeat(np.repeat(flat, 4, axis=0), 4, axis=1)[:height, :width].copy()
    for r in sorted(set(int(v) for v in noisy_block_rows)):
        if 0 <= r < nbh:
            y0, y1 = r * 4, min(height, r * 4 + 4)
            out[y0:y1] = survey_rgba8(width, 4, seed + r, alpha)[:y1 - y0]
    return out
