"""The integer (UINT / SINT), extended-range (R10G10B10_XR_BIAS_A2_UNORM) and 4:4:4 video (AYUV, Y410, Y416) formats of LoadScanline /
StoreScanline (DirectXTexConvert.cpp:805-842, :900-904, :913-981, :1031-1156, :1291-1394 and :1674-1850, :1873-2016, :2173-2272):
the HIP kernels against the reference's own scanline layer (DirectXTexConvert.cpp compiled in place into oracle/_ref over the DirectXMath
leaf shim, oracle/shim), and - on the CPU - that layer against an independent, vectorised statement in numpy (below). DirectXMath's integer
loads / stores are stated from its SSE2 paths in the shim (and checked against the x86 instructions by oracle/checks/shim_sse_check.cpp); the single-channel 8- / 16-bit integer formats and the video formats are the reference's own scalar code."""
import numpy as np
import pytest

UINTS = {3: (4, 32), 7: (3, 32), 12: (4, 16), 17: (2, 32), 30: (4, 8), 36: (2, 16), 42: (1, 32), 50: (2, 8), 57: (1, 16), 62: (1, 8)}
SINTS = {4: (4, 32), 8: (3, 32), 14: (4, 16), 18: (2, 32), 32: (4, 8), 38: (2, 16), 43: (1, 32), 52: (2, 8), 59: (1, 16), 64: (1, 8)}
RGB10A2_UINT, XR_BIAS, AYUV, Y410, Y416 = 25, 89, 100, 101, 102
NEW = sorted(UINTS) + sorted(SINTS) + [RGB10A2_UINT, XR_BIAS, AYUV, Y410, Y416]
RGBA32F, RGBA16F, RGBA8, RGBA8S = 2, 10, 28, 31
F = np.float32


# ---- numpy statement of the same loads / stores -------------------------------------------------------------------------------------
def np_load(raw, fmt, n):
    """n texels of `fmt` -> (n, 4) float32 as LoadScanline leaves them in the row buffer."""
    out = np.zeros((n, 4), F); out[:, 3] = 1
    if fmt in UINTS or fmt in SINTS:
        ch, bits = (UINTS if fmt in UINTS else SINTS)[fmt]
        signed = fmt in SINTS
        dt = {8: np.int8 if signed else np.uint8, 16: np.int16 if signed else np.uint16, 32: np.int32 if signed else np.uint32}[bits]
        v = raw.view(dt).reshape(n, ch)
        if bits == 32 and not signed:
            lo = (v & np.uint32(0x7FFFFFFF)).astype(np.int32).astype(F)          # cvtdq2ps of the low 31 bits ...
            f = np.where(v >> 31, (lo + F(2147483648.0)).astype(F), lo)           # ... + 2^31
        else:
            f = v.astype(F)
        out[:, :ch] = f
        return out
    w = raw.view(np.uint32).reshape(n, -1) if fmt != Y416 else None
    if fmt == RGB10A2_UINT:
        u = w[:, 0]
        return np.stack([(u & 1023).astype(F), ((u >> 10) & 1023).astype(F), ((u >> 20) & 1023).astype(F), (u >> 30).astype(F)], 1)
    if fmt == XR_BIAS:
        u = w[:, 0]
        c = lambda s: (((u >> s) & 1023).astype(np.int32) - 0x180).astype(F) * (F(1.0) / F(510.0))       # the SSE2 path: a multiplication by float(1/510)
        return np.stack([c(0), c(10), c(20), (u >> 30).astype(F) * (F(1.0) / F(3.0))], 1)
    if fmt == AYUV:
        b = raw.reshape(n, 4).astype(np.int64)
        v, u, y, a = b[:, 0] - 128, b[:, 1] - 128, b[:, 2] - 16, b[:, 3]
        r, g, bl = (298 * y + 409 * v + 128) >> 8, (298 * y - 100 * u - 208 * v + 128) >> 8, (298 * y + 516 * u + 128) >> 8
        q = lambda t: np.clip(t, 0, 255).astype(F) / F(255.0)
        return np.stack([q(r), q(g), q(bl), a.astype(F) / F(255.0)], 1)
    if fmt == Y410:
        x = w[:, 0].astype(np.int64)
        u, y, v, a = (x & 1023) - 512, ((x >> 10) & 1023) - 64, ((x >> 20) & 1023) - 512, x >> 30
        r, g, bl = (76533 * y + 104905 * v + 32768) >> 16, (76533 * y - 25747 * u - 53425 * v + 32768) >> 16, (76533 * y + 132590 * u + 32768) >> 16
        q = lambda t: np.clip(t, 0, 1023).astype(F) / F(1023.0)
        return np.stack([q(r), q(g), q(bl), a.astype(F) / F(3.0)], 1)
    if fmt == Y416:
        h = raw.view(np.uint16).reshape(n, 4).astype(np.int64)
        u, y, v, a = h[:, 0] - 32768, h[:, 1] - 4096, h[:, 2] - 32768, h[:, 3]
        r, g, bl = (76607 * y + 105006 * v + 32768) >> 16, (76607 * y - 25772 * u - 53477 * v + 32768) >> 16, (76607 * y + 132718 * u + 32768) >> 16
        q = lambda t: np.clip(t, 0, 65535).astype(F) / F(65535.0)
        return np.stack([q(r), q(g), q(bl), q(a)], 1)
    raise NotImplementedError(fmt)


def _sse_clamp(v, lo, hi):
    s = np.where(v > F(lo), v, F(lo))                 # maxps(v, lo): NaN -> lo
    return np.where(s < F(hi), s, F(hi)).astype(F)


def np_store(v, fmt):
    """(n, 4) float32 -> bytes as StoreScanline writes them."""
    v = np.ascontiguousarray(v, F); n = v.shape[0]
    with np.errstate(invalid="ignore", over="ignore"):
        if fmt in UINTS or fmt in SINTS:
            ch, bits = (UINTS if fmt in UINTS else SINTS)[fmt]
            signed = fmt in SINTS
            x = v[:, :ch]
            if bits == 32 and not signed:
                s = _sse_clamp(x, 0.0, np.inf)
                big = s >= F(2147483648.0)
                t = np.where(big, (s - F(2147483648.0)).astype(F), s)
                q = np.trunc(np.where(s > F(4294967040.0), 0, t)).astype(np.int64).astype(np.uint32) ^ np.where(big, np.uint32(0x80000000), np.uint32(0))
                q = np.where(s > F(4294967040.0), np.uint32(0xFFFFFFFF), q)
                return q.astype(np.uint32).reshape(-1).view(np.uint8)
            if bits == 32:
                ok = (x >= F(-2147483648.0)) & ~(x > F(2147483520.0))
                q = np.where(ok, np.trunc(np.where(ok, x, 0)).astype(np.int64), np.where(x > F(2147483520.0), 0x7FFFFFFF, -0x80000000))
                return q.astype(np.int32).reshape(-1).view(np.uint8)
            hi = {(16, False): 65535.0, (16, True): 32767.0, (8, False): 255.0, (8, True): 127.0}[(bits, signed)]
            lo = -hi if signed else 0.0
            if ch == 1:         # the reference's scalar code: std::min / std::max, C++ cast (a NaN ends up 0)
                s = np.where(F(hi) < x, F(hi), x); s = np.where(s < F(lo), F(lo), s)
                q = np.trunc(np.nan_to_num(s, nan=0.0)).astype(np.int64)
            else:               # DirectXMath: maxps / minps, cvtps2dq (round to nearest even)
                q = np.rint(_sse_clamp(x, lo, hi)).astype(np.int64)
            dt = {8: np.uint8, 16: np.uint16}[bits]
            return (q & (0xFF if bits == 8 else 0xFFFF)).astype(dt).reshape(-1).view(np.uint8)
        if fmt == RGB10A2_UINT:
            q = lambda c, hi: np.trunc(_sse_clamp(v[:, c], 0.0, hi)).astype(np.uint32)
            return (q(0, 1023.0) | (q(1, 1023.0) << 10) | (q(2, 1023.0) << 20) | (q(3, 3.0) << 30)).astype(np.uint32).view(np.uint8)
        if fmt == XR_BIAS:
            def q(c, scale, bias, hi):
                t = ((v[:, c] * F(scale)).astype(F) + F(bias)).astype(F)
                return np.trunc(_sse_clamp(t, 0.0, hi)).astype(np.uint32)
            return ((q(0, 510, 384, 1023) & 1023) | ((q(1, 510, 384, 1023) & 1023) << 10) | ((q(2, 510, 384, 1023) & 1023) << 20) | (q(3, 3, 0, 3) << 30)).astype(np.uint32).view(np.uint8)
        sat = lambda c, scale: np.trunc((_sse_clamp(v[:, c], 0.0, 1.0) * F(scale)).astype(F)).astype(np.int64)
        if fmt == AYUV:
            r, g, b, a = sat(0, 255), sat(1, 255), sat(2, 255), sat(3, 255)
            y, u, w = ((66 * r + 129 * g + 25 * b + 128) >> 8) + 16, ((-38 * r - 74 * g + 112 * b + 128) >> 8) + 128, ((112 * r - 94 * g - 18 * b + 128) >> 8) + 128
            return np.stack([np.clip(w, 0, 255), np.clip(u, 0, 255), np.clip(y, 0, 255), a], 1).astype(np.uint8).reshape(-1)
        if fmt == Y410:
            r, g, b, a = sat(0, 1023), sat(1, 1023), sat(2, 1023), sat(3, 3)
            y, u, w = ((16780 * r + 32942 * g + 6544 * b + 32768) >> 16) + 64, ((-9683 * r - 19017 * g + 28700 * b + 32768) >> 16) + 512, ((28700 * r - 24033 * g - 4667 * b + 32768) >> 16) + 512
            return (np.clip(u, 0, 1023) | (np.clip(y, 0, 1023) << 10) | (np.clip(w, 0, 1023) << 20) | (a << 30)).astype(np.uint32).view(np.uint8)
        if fmt == Y416:
            usn = lambda c: np.rint((_sse_clamp(v[:, c], 0.0, 1.0) * F(65535.0)).astype(F)).astype(np.int64)
            r, g, b, a = usn(0), usn(1), usn(2), usn(3)
            y, u, w = ((16763 * r + 32910 * g + 6537 * b + 32768) >> 16) + 4096, ((-9674 * r - 18998 * g + 28672 * b + 32768) >> 16) + 32768, ((28672 * r - 24010 * g - 4662 * b + 32768) >> 16) + 32768
            return np.stack([np.clip(u, 0, 65535), np.clip(y, 0, 65535), np.clip(w, 0, 65535), a], 1).astype(np.uint16).reshape(-1).view(np.uint8)
    raise NotImplementedError(fmt)


def _values(rng, n):
    """Floats around every rounding / clamping boundary of the integer stores."""
    v = (rng.random((n, 4), dtype=F) * 600 - 200).astype(F)
    scale = np.exp2(rng.integers(-4, 34, (n, 1))).astype(F)
    v = np.where(rng.random((n, 1)) < 0.5, (v * scale).astype(F), v)
    edge = np.array([0, 0.5, 1.5, 2.5, -0.5, -1.5, 126.5, 127, 127.5, 128, 254.5, 255, 255.5, 256, 32766.5, 32767, 32767.5, 32768, 65534.5, 65535, 65535.5, 65536,
                     1022.5, 1023, 1023.9, 1024, 2.9, 3, 3.5, 2147483520, 2147483648, 2147483904, 4294967040, 4294967296, 8e9, -2147483648, -2147483904, -9e9, -127.5, -128,
                     -32767.5, -32768, 0.9999999, 1.0000001, -0.7529412, 1.2529413, 0.25, 0.75], F)
    k = min(n, edge.size)
    v[:k, 0] = edge[:k]; v[:k, 1] = edge[:k][::-1]; v[:k, 2] = -edge[:k]; v[:k, 3] = edge[:k] * F(0.5)
    return v


@pytest.mark.parametrize("fmt", NEW)
def test_numpy_statement_agrees_with_the_reference_layer(oracle, fmt):
    """CPU: load every bit pattern (random) and store boundary values through the reference's Convert (ConvertCustom's loop with fp32 on
    the other side, so ConvertScanline has nothing to do to the values except the UNORM saturation the float -> UNORM branch applies) and
    through the numpy statement above."""
    w, h = 97, 3
    rng = np.random.default_rng(fmt)
    raw = rng.integers(0, 256, oracle.image_bytes(fmt, w, h), dtype=np.uint8)
    got = oracle.ref_convert(raw, w, h, fmt, RGBA32F, 0, 0.5).view(F).reshape(-1, 4)
    want = np_load(raw, fmt, w * h)
    if (UINTS.get(fmt) or SINTS.get(fmt) or (4, 0))[0] == 1:
        want[:, 1] = want[:, 0]; want[:, 2] = want[:, 0]              # R -> RGB formats: ConvertScanline replicates red (:3667-3679)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), fmt
    v = _values(rng, w * h)
    ref = oracle.ref_convert(v, w, h, RGBA32F, fmt, 0, 0.5)
    is_unorm = fmt in (XR_BIAS, AYUV, Y410, Y416)
    x = np.where(v > 0, v, F(0)); x = np.where(x < 1, x, F(1))            # FLOAT -> UNORM: XMVectorSaturate (:3481-3486)
    mine = np_store(x.astype(F) if is_unorm else v, fmt)
    assert np.array_equal(ref, mine), (fmt, np.nonzero(ref != mine)[0][:8])


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", NEW)
@pytest.mark.parametrize("other", [RGBA32F, RGBA16F, RGBA8, RGBA8S])
def test_convert_int_yuv(ctx, oracle, fmt, other):
    """Every new format as the source and as the destination of Convert against float, half, UNORM and SNORM four-channel formats."""
    w, h = 67, 9
    rng = np.random.default_rng(fmt * 31 + other)
    raw = rng.integers(0, 256, oracle.image_bytes(fmt, w, h), dtype=np.uint8)
    for flags in (0, 0x2000):            # default and TEX_FILTER_RGB_COPY_GREEN (the channel-count branches)
        got = ctx.convert(raw, w, h, fmt, other, flags, 0.5)
        ref = oracle.ref_convert(raw, w, h, fmt, other, flags, 0.5)
        assert np.array_equal(got, ref), ("load", fmt, other, hex(flags), np.nonzero(got != ref)[0][:8])
    if other in (RGBA32F, RGBA16F):
        v = _values(rng, w * h)
        src = v if other == RGBA32F else np.clip(v, -65000, 65000).astype(np.float16)
    else:
        src = rng.integers(0, 256, oracle.image_bytes(other, w, h), dtype=np.uint8)
    got = ctx.convert(src, w, h, other, fmt, 0, 0.5)
    ref = oracle.ref_convert(src, w, h, other, fmt, 0, 0.5)
    assert np.array_equal(got, ref), ("store", other, fmt, np.nonzero(got != ref)[0][:8])


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [3, 14, 30, 32, 36, 42, 59, 62, RGB10A2_UINT, XR_BIAS, AYUV, Y410, Y416])
@pytest.mark.parametrize("flt", [0x400000, 0x300000, 0x500000])       # box, cubic, triangle
def test_mips_and_resize_int_yuv(ctx, oracle, fmt, flt):
    """The filters read and write the new formats through the same Load / StoreScanline code (mip chain + an arbitrary-ratio resize)."""
    w, h = 32, 16
    rng = np.random.default_rng(fmt + flt)
    img = rng.integers(0, 256, oracle.image_bytes(fmt, w, h), dtype=np.uint8)
    if fmt in (3, 42):
        img = (img.view(np.uint32) >> np.uint32(rng.integers(0, 24))).astype(np.uint32).view(np.uint8)      # keep sums of four inside fp32-exact territory some of the time
    got = ctx.generate_mips(img, w, h, fmt, 5, flt)
    ref = oracle.ref_generate_mips(img, w, h, fmt, flt, 5)
    for lvl in range(5):
        assert np.array_equal(got[lvl], ref[lvl]), (fmt, hex(flt), lvl)
    if flt != 0x400000:
        assert np.array_equal(ctx.resize(img, w, h, fmt, 21, 13, flt), oracle.ref_resize(img, w, h, fmt, 21, 13, flt)), (fmt, hex(flt))


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [12, 30, 32, XR_BIAS, AYUV, Y410, Y416])
@pytest.mark.parametrize("bc", [71, 77, 83, 98])
def test_compress_from_int_yuv(ctx, oracle, fmt, bc):
    """Compress takes them as sources through the same tile loader (DirectXTexCompress.cpp:210-372 -> LoadScanline -> ConvertScanline)."""
    w, h = 36, 20
    rng = np.random.default_rng(fmt * 7 + bc)
    img = rng.integers(0, 256, oracle.image_bytes(fmt, w, h), dtype=np.uint8)
    assert np.array_equal(ctx.compress(img, w, h, fmt, bc, 0, 0.5), oracle.ref_compress_image(img, w, h, fmt, bc, 0, 0.5)), (fmt, bc)
