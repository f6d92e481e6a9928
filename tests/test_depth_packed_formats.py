"""Depth / stencil formats (D32_FLOAT_S8X24_UINT, D32_FLOAT, D24_UNORM_S8_UINT, D16_UNORM) of LoadScanline / StoreScanline
(DirectXTexConvert.cpp:844-860, :937, :982-997, :1053 and :1725-1744, :1810, :1852-1869, :1897) and ConvertScanline's depth branch
(:3186-3451): the HIP kernels against the reference's own scanline layer (DirectXTexConvert.cpp compiled in place into oracle/_ref), and - on
the CPU - that layer against an independent statement in numpy (below). These cases are the reference's own scalar code (no DirectXMath packed
type), except the XMVectorClamp / XMVectorSaturate / XMVectorMultiplyAdd steps (maxps / minps, unfused multiply-add)."""
import numpy as np
import pytest

D32S8, D32, D24S8, D16 = 20, 40, 45, 55
DEPTH = [D32S8, D32, D24S8, D16]
RGBA32F, RGBA16F, RGBA8, RGBA8S, R32F, R16U = 2, 10, 28, 31, 41, 56
F = np.float32
IS_FLOAT_DEPTH = {D32S8: True, D32: True, D24S8: False, D16: False}
HAS_STENCIL = {D32S8: True, D32: False, D24S8: True, D16: False}


def _clamp(v, lo, hi):
    s = np.where(v > F(lo), v, F(lo))                 # maxps(v, lo): NaN -> lo
    return np.where(s < F(hi), s, F(hi)).astype(F)


def np_load(raw, fmt, n):
    """n texels -> (n, 4) float32 rows as LoadScanline leaves them: (depth, stencil, 0, 1)."""
    out = np.zeros((n, 4), F); out[:, 3] = 1
    if fmt == D32:
        out[:, 0] = raw.view(F)
    elif fmt == D32S8:
        w = raw.view(np.uint32).reshape(n, 2)
        out[:, 0] = w[:, 0].view(F); out[:, 1] = (w[:, 1] & 0xFF).astype(F)
    elif fmt == D24S8:
        w = raw.view(np.uint32)
        out[:, 0] = (w & 0xFFFFFF).astype(F) / F(16777215.0); out[:, 1] = (w >> 24).astype(F)
    else:
        out[:, 0] = raw.view(np.uint16).astype(F) / F(65535.0)
    return out


def np_store(v, fmt):
    v = np.ascontiguousarray(v, F); n = v.shape[0]
    with np.errstate(invalid="ignore", over="ignore"):
        if fmt == D32:
            return v[:, 0].copy().view(np.uint8)
        if fmt == D32S8:
            s = np.where(F(0) < v[:, 1], v[:, 1], F(0)); s = np.where(s < F(255), s, F(255))       # std::min(255, std::max(0, y)): NaN -> 0
            out = np.zeros((n, 2), np.uint32)
            out[:, 0] = v[:, 0].copy().view(np.uint32); out[:, 1] = np.trunc(s).astype(np.uint32) & 0xFF
            return out.reshape(-1).view(np.uint8)
        if fmt == D24S8:
            d = np.trunc((_clamp(v[:, 0], 0, 1) * F(16777215.0)).astype(F)).astype(np.uint32) & 0xFFFFFF
            s = np.trunc(_clamp(v[:, 1], 0, 255)).astype(np.uint32) & 0xFF
            return (d | (s << 24)).astype(np.uint32).view(np.uint8)
        x = np.where(v[:, 0] < F(1), v[:, 0], F(1)); x = np.where(x > F(0), x, F(0))               # std::max(std::min(v, 1), 0)
        return np.trunc(((np.nan_to_num(x, nan=0.0) * F(65535.0)).astype(F) + F(0.5)).astype(F)).astype(np.uint16).view(np.uint8)


def np_depth_to_float4(rows, fmt):
    """ConvertScanline depth -> R32G32B32A32_FLOAT (:3189-3291): stencil to alpha as it is, depth splat to RGB."""
    out = rows.copy()
    if HAS_STENCIL[fmt]:
        out[:, 3] = rows[:, 1]
    out[:, 1] = rows[:, 0]; out[:, 2] = rows[:, 0]
    return out


def np_float4_to_depth(v, fmt):
    """ConvertScanline R32G32B32A32_FLOAT -> depth (:3293-3434): red to depth, saturated for a UNORM depth, alpha to stencil as it is."""
    out = v.copy()
    if not IS_FLOAT_DEPTH[fmt]:
        out[:, 0] = _clamp(v[:, 0], 0, 1)
    if HAS_STENCIL[fmt]:
        out[:, 1] = v[:, 3]
    return out


def _values(rng, n):
    v = (rng.random((n, 4), dtype=F) * 3 - 1).astype(F)
    v[:, 3] = (rng.random(n, dtype=F) * 300 - 20).astype(F)                 # alpha -> stencil: around 0 and 255
    edge = np.array([0, 1, 0.5, -0.25, 1.25, 0.9999999, 1.0000001, 5.9604645e-08, 2.9802322e-08, 0.99999994, 254.5, 255, 255.5, 256, -1, 1e-30, 7.62951e-06, 1.52590e-05], F)
    k = min(n, edge.size)
    v[:k, 0] = edge[:k]; v[:k, 3] = edge[:k][::-1]
    return v


@pytest.mark.parametrize("fmt", DEPTH)
def test_numpy_statement_agrees_with_the_reference_layer(oracle, fmt):
    w, h = 97, 3
    rng = np.random.default_rng(fmt)
    raw = rng.integers(0, 256, oracle.image_bytes(fmt, w, h), dtype=np.uint8)
    if fmt in (D32, D32S8):                            # keep the float depths finite
        words = raw.view(np.uint32).reshape(w * h, -1)
        words[:, 0] = (rng.random(w * h, dtype=F) * 4 - 1.5).astype(F).view(np.uint32)
    got = oracle.ref_convert(raw, w, h, fmt, RGBA32F, 0, 0.5).view(F).reshape(-1, 4)
    want = np_depth_to_float4(np_load(raw, fmt, w * h), fmt)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), fmt
    v = _values(rng, w * h)
    ref = oracle.ref_convert(v, w, h, RGBA32F, fmt, 0, 0.5)
    mine = np_store(np_float4_to_depth(v, fmt), fmt)
    assert np.array_equal(ref, mine), (fmt, np.nonzero(ref != mine)[0][:8])


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", DEPTH)
@pytest.mark.parametrize("other", [RGBA32F, RGBA16F, RGBA8, RGBA8S, R32F, R16U] + DEPTH)
def test_convert_depth(ctx, oracle, fmt, other):
    """Every depth format as the source and as the destination of Convert: float, half, UNORM and SNORM colour formats (all the stencil <->
    alpha and depth <-> colour steps), single-channel formats, and the other depth formats (float depth -> UNORM depth saturates)."""
    if other == fmt:
        pytest.skip("Convert refuses identical formats (DirectXTexConvert.cpp:4804-4813)")
    w, h = 67, 9
    rng = np.random.default_rng(fmt * 31 + other)
    raw = rng.integers(0, 256, oracle.image_bytes(fmt, w, h), dtype=np.uint8)
    if fmt in (D32, D32S8):
        words = raw.view(np.uint32).reshape(w * h, -1)
        words[:, 0] = (rng.random(w * h, dtype=F) * 4 - 1.5).astype(F).view(np.uint32)
    for flags in (0, 0x2000, 0x8000):            # default, TEX_FILTER_RGB_COPY_GREEN, TEX_FILTER_RGB_COPY_ALPHA
        got = ctx.convert(raw, w, h, fmt, other, flags, 0.5)
        ref = oracle.ref_convert(raw, w, h, fmt, other, flags, 0.5)
        assert np.array_equal(got, ref), ("from depth", fmt, other, hex(flags), np.nonzero(got != ref)[0][:8])
    if other == RGBA32F:
        src = _values(rng, w * h)
    elif other == RGBA16F:
        src = np.clip(_values(rng, w * h), -65000, 65000).astype(np.float16)
    elif other == R32F:
        src = _values(rng, w * h)[:, 0].copy()
    else:
        src = rng.integers(0, 256, oracle.image_bytes(other, w, h), dtype=np.uint8)
        if other in (D32, D32S8):
            words = src.view(np.uint32).reshape(w * h, -1)
            words[:, 0] = (rng.random(w * h, dtype=F) * 4 - 1.5).astype(F).view(np.uint32)
    for flags in (0, 0x2000, 0x4000, 0x8000):    # + COPY_BLUE: which channel becomes the depth
        got = ctx.convert(src, w, h, other, fmt, flags, 0.5)
        ref = oracle.ref_convert(src, w, h, other, fmt, flags, 0.5)
        assert np.array_equal(got, ref), ("to depth", other, fmt, hex(flags), np.nonzero(got != ref)[0][:8])


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", DEPTH)
@pytest.mark.parametrize("flt", [0x400000, 0x300000, 0x500000])       # box, cubic, triangle
def test_mips_and_resize_depth(ctx, oracle, fmt, flt):
    w, h = 32, 16
    rng = np.random.default_rng(fmt + flt)
    img = rng.integers(0, 256, oracle.image_bytes(fmt, w, h), dtype=np.uint8)
    if fmt in (D32, D32S8):
        words = img.view(np.uint32).reshape(w * h, -1)
        words[:, 0] = rng.random(w * h, dtype=F).view(np.uint32)
    got = ctx.generate_mips(img, w, h, fmt, 5, flt)
    ref = oracle.ref_generate_mips(img, w, h, fmt, flt, 5)
    for lvl in range(5):
        assert np.array_equal(got[lvl], ref[lvl]), (fmt, hex(flt), lvl)
    if flt != 0x400000:
        assert np.array_equal(ctx.resize(img, w, h, fmt, 21, 13, flt), oracle.ref_resize(img, w, h, fmt, 21, 13, flt)), (fmt, hex(flt))


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", DEPTH)
@pytest.mark.parametrize("bc", [71, 77, 81, 83, 95, 98])
def test_compress_from_depth(ctx, oracle, fmt, bc):
    """Compress takes them as sources through the tile loader: depth splat to RGB (saturated / re-ranged for the UNORM / SNORM codecs),
    stencil to alpha (DirectXTexCompress.cpp:210-372 -> LoadScanline -> ConvertScanline's depth branch)."""
    w, h = 36, 20
    rng = np.random.default_rng(fmt * 7 + bc)
    img = rng.integers(0, 256, oracle.image_bytes(fmt, w, h), dtype=np.uint8)
    if fmt in (D32, D32S8):
        words = img.view(np.uint32).reshape(w * h, -1)
        words[:, 0] = (rng.random(w * h, dtype=F) * 1.5 - 0.25).astype(F).view(np.uint32)
    assert np.array_equal(ctx.compress(img, w, h, fmt, bc, 0, 0.5), oracle.ref_compress_image(img, w, h, fmt, bc, 0, 0.5)), (fmt, bc)


# ---- formats whose memory element holds several texels: R1_UNORM (8 per byte), R8G8_B8G8 / G8R8_G8B8 / YUY2 (2 per dword), Y210 / Y216 (2 per
# qword): LoadScanline :1171-1226, :1399-1510; StoreScanline :2033-2094, :2274-2397 ---------------------------------------------------------
R1, RGBG, GRGB, YUY2, Y210, Y216 = 66, 68, 69, 107, 108, 109
GROUPED = [R1, RGBG, GRGB, YUY2, Y210, Y216]


def _row_elems(fmt, w):
    return (w + 7) // 8 if fmt == R1 else (w + 1) // 2


def np_load_group(raw, fmt, w, h):
    """-> (h, w, 4) float32 rows as LoadScanline leaves them."""
    ne = _row_elems(fmt, w)
    out = np.zeros((h, ne * (8 if fmt == R1 else 2), 4), F); out[..., 3] = 1
    if fmt == R1:
        b = raw.reshape(h, ne)
        bits = (b[:, :, None] >> np.arange(7, -1, -1)[None, None, :]) & 1
        out[..., 0] = bits.reshape(h, ne * 8).astype(F)
        return out[:, :w]
    if fmt in (RGBG, GRGB):
        v = raw.reshape(h, ne, 4).astype(F) * F(1.0 / 255.0)
        if fmt == RGBG:
            r, g0, b, g1 = v[..., 0], v[..., 1], v[..., 2], v[..., 3]
        else:
            g0, r, g1, b = v[..., 0], v[..., 1], v[..., 2], v[..., 3]
        out[:, 0::2, 0] = r; out[:, 1::2, 0] = r; out[:, 0::2, 1] = g0; out[:, 1::2, 1] = g1; out[:, 0::2, 2] = b; out[:, 1::2, 2] = b
        return out[:, :w]
    if fmt == YUY2:
        e = raw.reshape(h, ne, 4).astype(np.int64)
        ys = (e[..., 0] - 16, e[..., 2] - 16); u = e[..., 1] - 128; v = e[..., 3] - 128
        coef = (298, 409, 100, 208, 516, 128, 8, 255)
    else:
        e = raw.view(np.uint16).reshape(h, ne, 4).astype(np.int64)
        if fmt == Y210:
            e = e >> 6
            ys = (e[..., 0] - 64, e[..., 2] - 64); u = e[..., 1] - 512; v = e[..., 3] - 512
            coef = (76533, 104905, 25747, 53425, 132590, 32768, 16, 1023)
        else:
            ys = (e[..., 0] - 4096, e[..., 2] - 4096); u = e[..., 1] - 32768; v = e[..., 3] - 32768
            coef = (76607, 105006, 25772, 53477, 132718, 32768, 16, 65535)
    cy, crv, cgu, cgv, cbu, rnd, sh, top = coef
    for k in range(2):
        y = ys[k]
        r = (cy * y + crv * v + rnd) >> sh; g = (cy * y - cgu * u - cgv * v + rnd) >> sh; b = (cy * y + cbu * u + rnd) >> sh
        for c, t in enumerate((r, g, b)):
            out[:, k::2, c] = np.clip(t, 0, top).astype(F) / F(top)
    return out[:, :w]


def np_store_group(v, fmt):
    """(h, w, 4) float32 -> bytes as StoreScanline writes the rows (a missing second texel is a zero vector)."""
    v = np.ascontiguousarray(v, F); h, w = v.shape[:2]
    ne = _row_elems(fmt, w)
    per = 8 if fmt == R1 else 2
    pad = np.zeros((h, ne * per, 4), F); pad[:, :w] = v
    with np.errstate(invalid="ignore", over="ignore"):
        if fmt == R1:
            bits = (pad[..., 0] > F(0.25)).reshape(h, ne, 8)
            return (bits * (1 << np.arange(7, -1, -1))[None, None, :]).sum(-1).astype(np.uint8).reshape(-1)
        t0, t1 = pad[:, 0::2], pad[:, 1::2]
        sat = lambda x: _clamp(x, 0, 1)
        if fmt in (RGBG, GRGB):
            q = lambda x: np.trunc((_clamp((x + F(0.5 / 255.0)).astype(F), 0, 1) * F(255.0)).astype(F)).astype(np.uint8)
            r, g0, b, g1 = q(t0[..., 0]), q(t0[..., 1]), q(t0[..., 2]), q(t1[..., 1])
            return (np.stack([r, g0, b, g1], -1) if fmt == RGBG else np.stack([g0, r, g1, b], -1)).reshape(-1)
        if fmt == YUY2:
            q = lambda x: np.trunc((sat(x) * F(255.0)).astype(F)).astype(np.int64)
            m = ((66, 129, 25, 16), (-38, -74, 112, 128), (112, -94, -18, 128)); rnd, sh, top, shl = 128, 8, 255, 0
        elif fmt == Y210:
            q = lambda x: np.trunc((sat(x) * F(1023.0)).astype(F)).astype(np.int64)
            m = ((16780, 32942, 6544, 64), (-9683, -19017, 28700, 512), (28700, -24033, -4667, 512)); rnd, sh, top, shl = 32768, 16, 1023, 6
        else:
            q = lambda x: np.rint((sat(x) * F(65535.0)).astype(F)).astype(np.int64)
            m = ((16763, 32910, 6537, 4096), (-9674, -18998, 28672, 32768), (28672, -24010, -4662, 32768)); rnd, sh, top, shl = 32768, 16, 65535, 0
        yuv = []
        for t in (t0, t1):
            r, g, b = q(t[..., 0]), q(t[..., 1]), q(t[..., 2])
            yuv.append([((a * r + bb * g + c * b + rnd) >> sh) + o for a, bb, c, o in m])
        y0, y1 = yuv[0][0], yuv[1][0]
        u = (yuv[0][1] + yuv[1][1]) >> 1; vv = (yuv[0][2] + yuv[1][2]) >> 1
        e = np.stack([np.clip(y0, 0, top), np.clip(u, 0, top), np.clip(y1, 0, top), np.clip(vv, 0, top)], -1) << shl
        return e.astype(np.uint8 if fmt == YUY2 else np.uint16).reshape(-1).view(np.uint8)


def _group_values(rng, w, h):
    v = (rng.random((h, w, 4), dtype=F) * 1.6 - 0.3).astype(F)
    edge = np.array([0, 1, 0.25, 0.25000003, 0.24999999, 0.5, -0.25, 1.25, 0.9999999, 1.0000001, 0.00196, 0.00197, 0.998, 0.99805, 127.5 / 255, 128 / 255], F)
    k = min(w, edge.size)
    v[0, :k, 0] = edge[:k]; v[0, :k, 1] = edge[:k][::-1]; v[0, :k, 2] = (1 - edge[:k]).astype(F)
    return v


@pytest.mark.parametrize("fmt", GROUPED)
@pytest.mark.parametrize("w", [97, 32, 1])
def test_numpy_statement_agrees_with_the_reference_layer_grouped(oracle, fmt, w):
    h = 3
    rng = np.random.default_rng(fmt * 7 + w)
    raw = rng.integers(0, 256, oracle.image_bytes(fmt, w, h), dtype=np.uint8)
    got = oracle.ref_convert(raw, w, h, fmt, RGBA32F, 0, 0.5).view(F).reshape(h, w, 4)
    want = np_load_group(raw, fmt, w, h)
    if fmt == R1:
        want[..., 1] = want[..., 0]; want[..., 2] = want[..., 0]            # R -> RGB formats: ConvertScanline replicates red (:3667-3679)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), fmt
    v = _group_values(rng, w, h)
    ref = oracle.ref_convert(v, w, h, RGBA32F, fmt, 0, 0.5)
    mine = np_store_group(_clamp(v, 0, 1), fmt)                            # FLOAT -> UNORM: XMVectorSaturate (:3481-3486)
    assert np.array_equal(ref, mine), (fmt, np.nonzero(ref != mine)[0][:8])


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", GROUPED)
@pytest.mark.parametrize("other", [RGBA32F, RGBA16F, RGBA8, RGBA8S, YUY2, RGBG, R1])
@pytest.mark.parametrize("w", [67, 64])
def test_convert_grouped(ctx, oracle, fmt, other, w):
    """Every grouped format as the source and as the destination of Convert (odd width: the last element holds one texel)."""
    if other == fmt:
        pytest.skip("Convert refuses identical formats")
    h = 9
    rng = np.random.default_rng(fmt * 31 + other + w)
    raw = rng.integers(0, 256, oracle.image_bytes(fmt, w, h), dtype=np.uint8)
    for flags in (0, 0x2000):
        got = ctx.convert(raw, w, h, fmt, other, flags, 0.5)
        ref = oracle.ref_convert(raw, w, h, fmt, other, flags, 0.5)
        assert np.array_equal(got, ref), ("from", fmt, other, hex(flags), np.nonzero(got != ref)[0][:8])
    if other == RGBA32F:
        src = _group_values(rng, w, h)
    elif other == RGBA16F:
        src = _group_values(rng, w, h).astype(np.float16)
    else:
        src = rng.integers(0, 256, oracle.image_bytes(other, w, h), dtype=np.uint8)
    got = ctx.convert(src, w, h, other, fmt, 0, 0.5)
    ref = oracle.ref_convert(src, w, h, other, fmt, 0, 0.5)
    assert np.array_equal(got, ref), ("to", other, fmt, np.nonzero(got != ref)[0][:8])


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", GROUPED)
@pytest.mark.parametrize("flt", [0x400000, 0x300000, 0x500000, 0x200000])       # box, cubic, triangle, linear
def test_mips_and_resize_grouped(ctx, oracle, fmt, flt):
    w, h = 32, 16
    rng = np.random.default_rng(fmt + flt)
    img = rng.integers(0, 256, oracle.image_bytes(fmt, w, h), dtype=np.uint8)
    got = ctx.generate_mips(img, w, h, fmt, 5, flt)
    ref = oracle.ref_generate_mips(img, w, h, fmt, flt, 5)
    for lvl in range(5):
        assert np.array_equal(got[lvl], ref[lvl]), (fmt, hex(flt), lvl)
    if flt != 0x400000:
        assert np.array_equal(ctx.resize(img, w, h, fmt, 21, 13, flt), oracle.ref_resize(img, w, h, fmt, 21, 13, flt)), (fmt, hex(flt))


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", GROUPED)
@pytest.mark.parametrize("bc", [71, 98])
def test_compress_refuses_grouped_sources(ctx, oracle, fmt, bc):
    """Compress: the reference refuses R1_UNORM (DirectXTexCompress.cpp:228-232); for the packed formats its block loop takes an element
    of two texels for one texel and reads past the rows (see compress_view in csrc/capi.cpp) - refused here with the same HRESULT."""
    from directxtex_amd.capi import DxtexError
    w, h = 36, 20
    img = np.zeros(oracle.image_bytes(fmt, w, h), np.uint8)
    with pytest.raises(DxtexError) as e:
        ctx.compress(img, w, h, fmt, bc, 0, 0.5)
    assert e.value.hresult & 0xFFFFFFFF == 0x80070032
    if fmt == R1:
        with pytest.raises(oracle.RefError) as r:
            oracle.ref_compress_image(img, w, h, fmt, bc, 0, 0.5)
        assert r.value.hresult == 0x80070032


# ---- WIN11_DXGI_FORMAT_A4B4G4R4_UNORM (191; :1527-1541, :2419-2437) and the full list of formats the filters may treat as sRGB (:2825-2849) ----
A4B4G4R4 = 191


@pytest.mark.gpu
@pytest.mark.parametrize("other", [RGBA32F, RGBA8, 115])
def test_convert_a4b4g4r4(ctx, oracle, other):
    w, h = 67, 9
    rng = np.random.default_rng(other)
    raw = rng.integers(0, 256, oracle.image_bytes(A4B4G4R4, w, h), dtype=np.uint8)
    got = ctx.convert(raw, w, h, A4B4G4R4, other, 0, 0.5)
    assert np.array_equal(got, oracle.ref_convert(raw, w, h, A4B4G4R4, other, 0, 0.5))
    src = rng.random((h, w, 4), dtype=F) if other == RGBA32F else rng.integers(0, 256, oracle.image_bytes(other, w, h), dtype=np.uint8)
    got = ctx.convert(src, w, h, other, A4B4G4R4, 0, 0.5)
    assert np.array_equal(got, oracle.ref_convert(src, w, h, other, A4B4G4R4, 0, 0.5))
    if other == RGBA32F:
        nib = raw.view(np.uint16).astype(np.uint32)
        want = np.stack([(nib >> 12) & 15, (nib >> 8) & 15, (nib >> 4) & 15, nib & 15], -1).astype(F) * F(1.0 / 15.0)      # third statement of the load
        assert np.array_equal(ctx.convert(raw, w, h, A4B4G4R4, RGBA32F, 0, 0.5).view(F).reshape(-1, 4), want)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [6, 24, 26, 67, RGBG, GRGB, 85, 86, 115, A4B4G4R4])
@pytest.mark.parametrize("flt", [0x400000, 0x200000])
def test_filters_in_linear_space_for_the_whole_srgb_list(ctx, oracle, fmt, flt):
    """TEX_FILTER_SRGB makes the filters convert to linear and back for every format of LoadScanlineLinear's list (:2825-2849), not only
    the common ones: the result must differ from the plain filter and follow the reference (pow() on the device and libm's powf differ
    in the last bit for a few values, as in test_premultiply_alpha)."""
    w, h = 32, 16
    rng = np.random.default_rng(fmt + flt)
    img = rng.integers(0, 256, oracle.image_bytes(fmt, w, h), dtype=np.uint8)
    if fmt in (6, 26, 67):
        img = rng.random(w * h * 3, dtype=F) if fmt == 6 else img
    plain = ctx.generate_mips(img, w, h, fmt, 3, flt)
    got = ctx.generate_mips(img, w, h, fmt, 3, flt | 0x3000000)
    ref = oracle.ref_generate_mips(img, w, h, fmt, flt | 0x3000000, 3)
    assert any((g != p).any() for g, p in zip(got[1:], plain[1:])), "the sRGB flag did nothing"
    for lvl in range(1, 3):
        g, r = np.asarray(got[lvl]).view(np.uint8), np.asarray(ref[lvl]).view(np.uint8)
        assert (g != r).mean() < 0.02, (fmt, hex(flt), lvl, (g != r).mean())
