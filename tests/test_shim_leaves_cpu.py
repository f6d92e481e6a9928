"""The DirectXMath leaf shim (oracle/shim) under the reference's scanline layer. Since round 6 DirectXTexConvert.cpp is compiled in
place into oracle/_ref/libdxtex_ref.so (oracle/ref_convert.cpp); what this repo still STATES is DirectXMath itself (absent from the
image). These tests pin what can be pinned without it:
  * every scalar stand-in of an x86 instruction sequence equals that sequence run with <emmintrin.h> on the host CPU
    (oracle/checks/shim_sse_check.cpp: maxps / minps operand order, cvtps2dq / cvttps2dq, XMVectorRound's 2^23 trick, the unsigned
    detours);
  * libdxtex_ref.so carries the reference's LoadScanline / StoreScanline / ConvertScanline / Convert (no restated C++ is left);
  * load -> store round trips through the reference's own case analysis are the identity for every format whose texels are exact in fp32
    (a property of the reference + shim pair that a wrong field position, reciprocal or rounding mode would break)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
F = np.float32


def test_shim_primitives_equal_the_sse2_instructions():
    exe = os.path.join(REF_DIR, "shim_sse_check")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/shim_sse_check not built (make -C oracle ref)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:]
    assert " 0 mismatches" in out.stdout


def test_no_restated_scanline_layer_is_left():
    assert not os.path.exists(os.path.join(ROOT, "oracle", "restate")), "oracle/restate/ must stay deleted: the scanline layer is the reference's"
    lib = os.path.join(REF_DIR, "libdxtex_ref.so")
    if not os.path.exists(lib):
        pytest.skip("oracle/_ref/libdxtex_ref.so not built")
    syms = subprocess.run(["nm", "-DC", "--defined-only", lib], capture_output=True, text=True).stdout
    for name in ("DirectX::Internal::LoadScanline(", "DirectX::Internal::StoreScanline(", "DirectX::Internal::ConvertScanline(", "DirectX::Internal::StoreScanlineDither(",
                 "DirectX::Convert(DirectX::Image const&", "DirectX::ConvertToSinglePlane(", "dxtex_ref_load_scanline", "dxtex_ref_store_scanline"):
        assert name in syms, name
    obj = os.path.join(REF_DIR, "obj", "ref_convert.o")
    if os.path.exists(obj):          # where the library was built here: the scanline symbols come from the reference's translation unit
        assert "LoadScanline" in subprocess.run(["nm", "-C", "--defined-only", obj], capture_output=True, text=True).stdout


def _lib(oracle):
    lib = oracle.dxtex_oracle._load_ref()
    lib.dxtex_ref_load_scanline.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
    lib.dxtex_ref_store_scanline.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_float]
    return lib


# format, bytes per element, texels per element; every bit pattern of these is exact in fp32 and survives load -> store
# (R9G9B9E5 and R11G11B10 are excluded: not every pattern is canonical; R10G10B10A2_UNORM and XR_BIAS are excluded because XMStoreUDecN4 /
# XMStoreUDecN4_XR TRUNCATE and the reference adds no bias there: some codes come back one lower - see test_truncating_ten_bit_stores;
# signed -128 / -32768 are written back as -127 / -32767 by DirectXMath's symmetric clamps and are kept out of the input)
EXACT = [(28, 4, 1), (87, 4, 1), (88, 4, 1), (11, 8, 1), (35, 4, 1), (56, 2, 1), (49, 2, 1), (61, 1, 1), (65, 1, 1), (25, 4, 1), (85, 2, 1), (86, 2, 1),
         (115, 2, 1), (191, 2, 1), (12, 8, 1), (14, 8, 1), (30, 4, 1), (32, 4, 1), (36, 4, 1), (38, 4, 1), (50, 2, 1), (52, 2, 1), (57, 2, 1), (59, 2, 1), (62, 1, 1),
         (64, 1, 1), (10, 8, 1), (34, 4, 1), (54, 2, 1), (2, 16, 1), (6, 12, 1), (16, 8, 1), (41, 4, 1), (55, 2, 1), (68, 4, 2), (69, 4, 2)]


@pytest.mark.parametrize("fmt,nbytes,group", EXACT)
def test_load_store_round_trip_is_the_identity(oracle, fmt, nbytes, group):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    lib = _lib(oracle)
    n = 4096
    rng = np.random.default_rng(fmt)
    raw = rng.integers(0, 256, n * nbytes, dtype=np.uint8)
    if fmt in (10, 34, 54):                 # halves: finite values only (NaN payloads are not preserved bit for bit by design)
        h = raw.view(np.uint16); h[(h & 0x7C00) == 0x7C00] &= 0x3FFF
    if fmt in (2, 6, 16, 41):
        f = raw.view(np.uint32); f[(f & 0x7F800000) == 0x7F800000] &= 0x3FFFFFFF
    if fmt in (32, 52, 64):
        raw[raw == 0x80] = 0x81
    if fmt in (14, 38, 59):
        h = raw.view(np.uint16); h[h == 0x8000] = 0x8001
    if fmt == 88:
        raw.reshape(-1, 4)[:, 3] = 255      # B8G8R8X8: X is stored from w = 1 (XMVectorPermute with g_XMIdentityR3, :2165)
    rgba = np.zeros((n * group, 4), F)
    assert lib.dxtex_ref_load_scanline(raw.ctypes.data, raw.size, fmt, rgba.ctypes.data, n * group) == 0
    back = np.zeros_like(raw)
    assert lib.dxtex_ref_store_scanline(back.ctypes.data, back.size, fmt, rgba.ctypes.data, n * group, 0.5) == 0
    assert np.array_equal(raw, back), (fmt, np.flatnonzero(raw != back)[:8])


def test_truncating_ten_bit_stores(oracle):
    """R10G10B10A2_UNORM: XMLoadUDecN4 multiplies by the constant 1/1023 and XMStoreUDecN4 truncates v * 1023 (cvttps2dq; the reference adds
    its 0.5/255 bias only on the 8-bit paths, DirectXTexConvert.cpp:1747-1748 vs :1767), so a load -> store round trip returns code c as
    trunc(fl(fl(c * fl(1/1023)) * 1023)): one lower wherever the double rounding lands below c. Stated here in numpy, independently."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    lib = _lib(oracle)
    codes = np.arange(1024, dtype=np.uint32)
    raw = (codes | (codes << 10) | (codes << 20) | ((codes & 3) << 30)).astype(np.uint32)
    rgba = np.zeros((1024, 4), F)
    assert lib.dxtex_ref_load_scanline(raw.ctypes.data, raw.nbytes, 24, rgba.ctypes.data, 1024) == 0
    want = (codes.astype(F) * F(1.0 / 1023.0)).astype(F)
    assert np.array_equal(rgba[:, 0], want) and np.array_equal(rgba[:, 2], want)
    back = np.zeros_like(raw)
    assert lib.dxtex_ref_store_scanline(back.ctypes.data, back.nbytes, 24, rgba.ctypes.data, 1024, 0.5) == 0
    expect = np.trunc((want * F(1023.0)).astype(F)).astype(np.uint32)
    assert np.array_equal(back & 0x3FF, expect) and np.array_equal((back >> 20) & 0x3FF, expect)
    assert int((expect != codes).sum()) > 0          # the loss is real (and the HIP path reproduces it: tests/test_scanline_parity.py)
