"""Compress from every supported source format (the ConvertScanline branches CompressBC reaches), row pitches wider than
the image, and the C ABI's error codes, against the reference's real Compress driver (oracle/_ref)."""
import ctypes

import numpy as np
import pytest

import directxtex_amd as dx

pytestmark = pytest.mark.gpu

SRC = [28, 29, 87, 88, 2, 10, 11, 31, 49, 51, 61, 63, 65, 41, 54, 16, 34, 35, 56,
       6, 13, 24, 26, 37, 58, 67, 85, 86, 115]      # + the packed long tail: RGB32F, 16-bit SNORM, 10:10:10:2, 11:11:10, 9:9:9:5, 5:6:5, 5:5:5:1, 4:4:4:4
DST = [71, 72, 77, 80, 81, 83, 84, 98, 99, 95]


def _pixels(oracle, fmt, w, h, seed):
    rng = np.random.default_rng(seed)
    if fmt == 26:      # R11G11B10_FLOAT: finite codes only (a NaN texel makes the encoders' comparisons order-dependent)
        v = rng.integers(0, 2 ** 32, (h, w), dtype=np.uint64).astype(np.uint32)
        return v & np.uint32(~((1 << 10) | (1 << 21) | (1 << 31)) & 0xFFFFFFFF)
    if fmt in (2, 6, 16, 41):
        n = {2: 4, 6: 3, 16: 2, 41: 1}[fmt]
        return (rng.random((h, w, n), dtype=np.float32) * 1.5 - 0.25).astype(np.float32)
    if fmt in (10, 34, 54):
        n = {10: 4, 34: 2, 54: 1}[fmt]
        return (rng.random((h, w, n), dtype=np.float32) * 1.5 - 0.25).astype(np.float16)
    return rng.integers(0, 256, oracle.image_bytes(fmt, w, h), dtype=np.uint8)


@pytest.mark.parametrize("src", SRC)
@pytest.mark.parametrize("dst", DST)
def test_source_formats(ctx, oracle, src, dst):
    w, h = 23, 10
    px = _pixels(oracle, src, w, h, src * 100 + dst)
    flags = 0x100000 if dst in (98, 99) else 0         # BC7_QUICK keeps the oracle fast
    got = ctx.compress(px, w, h, src, dst, flags, 0.5)
    ref = oracle.ref_compress_image(px, w, h, src, dst, flags, 0.5)
    if px.dtype == np.uint8 or (src == 29) == (dst in (72, 99)):
        assert np.array_equal(got, ref), (src, dst, np.nonzero(got != ref)[0][:8])
    else:
        # one-sided sRGB on arbitrary floats goes through pow(): correctly rounded here, libm's powf in the reference,
        # which differ by 1 ulp on < 0.1 % of values; a block only changes when that ulp crosses a quantisation step
        bs = 8 if dst in (71, 72, 80, 81) else 16
        same = (got.reshape(-1, bs) == ref.reshape(-1, bs)).all(axis=1).mean()
        assert same >= 0.9, (src, dst, same)


@pytest.mark.parametrize("flags", [0x1000000, 0x2000000, 0x3000000])
@pytest.mark.parametrize("dst", [71, 77, 98, 95])
def test_srgb_flags(ctx, oracle, dst, flags):
    """TEX_COMPRESS_SRGB_IN / _OUT / both on 8-bit sources (DirectXTexCompress.cpp:34-45, DirectXTexConvert.cpp:3164-3180,3843-3853)."""
    w, h = 32, 16
    px = _pixels(oracle, 28, w, h, dst + (flags >> 24))
    f = flags | (0x100000 if dst == 98 else 0)
    src = 10 if dst == 95 else 28
    if src == 10:
        px = (np.random.default_rng(4).random((h, w, 4), dtype=np.float32)).astype(np.float16)
    got = ctx.compress(px, w, h, src, dst, f, 0.5)
    ref = oracle.ref_compress_image(px, w, h, src, dst, f, 0.5)
    if src == 28 or flags == 0x3000000:
        assert np.array_equal(got, ref)
    else:
        assert (got.reshape(-1, 16) == ref.reshape(-1, 16)).all(axis=1).mean() >= 0.9


def test_wide_row_pitch(ctx, oracle):
    w, h, pitch = 30, 18, 256
    rng = np.random.default_rng(3)
    buf = rng.integers(0, 256, (h, pitch), dtype=np.uint8)
    got = ctx.compress(buf, w, h, 28, 77, 0, 0.5, src_row_pitch=pitch)
    tight = np.ascontiguousarray(buf[:, :w * 4])
    assert np.array_equal(got, oracle.ref_compress_image(tight, w, h, 28, 77, 0, 0.5))
    assert np.array_equal(got, oracle.ref_compress_image(buf, w, h, 28, 77, 0, 0.5, row_pitch=pitch))


def test_c_abi_error_codes(ctx):
    from directxtex_amd import capi
    lib = capi._lib
    px = np.zeros(16 * 16 * 4, np.uint8); out = np.zeros(16 * 16, np.uint8)
    def img(arr, w, h, fmt):
        rp, sp = dx.compute_pitch(fmt, w, h)
        return capi.Image(w, h, fmt, rp, sp, arr.ctypes.data if arr is not None else None)
    h = ctx._h
    E_POINTER, E_INVALIDARG, E_FAIL, NOT_SUPPORTED = 0x80004003, 0x80070057, 0x80004005, 0x80070032
    u = lambda hr: hr & 0xFFFFFFFF
    s, d = img(px, 16, 16, 28), img(out, 16, 16, 77)
    assert u(lib.dxtex_compress(None, ctypes.byref(s), ctypes.byref(d), 0, 0.5)) == E_POINTER
    assert u(lib.dxtex_compress(h, ctypes.byref(img(None, 16, 16, 28)), ctypes.byref(d), 0, 0.5)) == E_POINTER      # null pixels, :80-81
    assert u(lib.dxtex_compress(h, ctypes.byref(d), ctypes.byref(d), 0, 0.5)) == E_INVALIDARG                        # compressed source, :671
    assert u(lib.dxtex_compress(h, ctypes.byref(s), ctypes.byref(s), 0, 0.5)) == E_INVALIDARG                        # uncompressed target
    assert u(lib.dxtex_compress(h, ctypes.byref(s), ctypes.byref(img(out, 12, 16, 77)), 0, 0.5)) == E_FAIL           # size mismatch, :800-804
    assert u(lib.dxtex_compress(h, ctypes.byref(capi.Image(16, 16, 27, 64, 1024, px.ctypes.data)), ctypes.byref(d), 0, 0.5)) == NOT_SUPPORTED     # R8G8B8A8_TYPELESS: HRESULT_E_NOT_SUPPORTED, :674-676
    assert u(lib.dxtex_decompress(h, ctypes.byref(s), ctypes.byref(s))) == E_INVALIDARG                              # :857
    assert b"" != lib.dxtex_ctx_last_error(h)
