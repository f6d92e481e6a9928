"""CPU-only checks of the test oracle itself (no GPU, no product code):
  * the reference has no golden vectors for this path, so the oracle is pinned the other way
    round: it IS the reference - BC.cpp / BC4BC5.cpp / BC6HBC7.cpp / DirectXTexCompress / Mipmaps / Resize / Misc.cpp
    and (round 6) DirectXTexConvert.cpp compiled in place into oracle/_ref - and these tests pin the thin layers around it (the numpy
    driver, the DirectXMath leaf shim) against each other and against format-level invariants;
  * the numpy restatement of the Compress driver (LoadScanline, tile gather, ConvertScanline) must agree byte for byte
    with the reference's real CompressBC driver running over the reference's own scanline layer."""
import numpy as np
import pytest

import oracle
from directxtex_amd import synth

RGBA8 = 28


def _need_ref():
    if not oracle.have_ref():
        pytest.fail("oracle/_ref/libdxtex_ref.so is missing: run `make -C oracle ref` where /root/reference exists")


@pytest.mark.parametrize("fmt", [71, 74, 77, 80, 81, 83, 84, 98])
@pytest.mark.parametrize("size", [(16, 16), (13, 7), (1, 1), (5, 18)])
def test_numpy_driver_equals_reference_driver(fmt, size):
    _need_ref()
    w, h = size
    img = synth.rgba8(w, h, seed=fmt + w, alpha="random")
    flags = 0x100000 if fmt == 98 else 0          # BC7_QUICK keeps the CPU suite short
    a = oracle.compress_image(img, w, h, RGBA8, fmt, flags, 0.5)
    b = oracle.ref_compress_image(img, w, h, RGBA8, fmt, flags, 0.5)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("fmt,dst", [(71, 28), (77, 28), (98, 28), (80, 61), (83, 49), (95, 2), (77, 10)])
def test_numpy_store_equals_reference_decompress(fmt, dst):
    _need_ref()
    rng = np.random.default_rng(fmt)
    w, h = 12, 9
    payload = rng.integers(0, 256, oracle.image_bytes(fmt, w, h), dtype=np.uint8)
    assert np.array_equal(oracle.decompress_image(payload, w, h, fmt, dst), oracle.ref_decompress_image(payload, w, h, fmt, dst))


def test_bc7_blocks_are_well_formed():
    """Structural self-checks from SURVEY.md section 8c: unary mode prefix, and decode(encode(x)) stays close to x."""
    _need_ref()
    rng = np.random.default_rng(1)
    tiles = np.clip(rng.random((64, 1, 4), dtype=np.float32) + rng.normal(0, 0.05, (64, 16, 4)).astype(np.float32), 0, 1)
    blocks = oracle.ref_encode_blocks(98, tiles, 0x100000)
    assert (blocks[:, 0] != 0).all()                       # a mode bit within the first byte
    dec = oracle.ref_decode_blocks(98, blocks)
    assert float(((dec - tiles) ** 2).mean()) < 2e-3


def test_bc1_roundtrip_flat_blocks_exact():
    _need_ref()
    vals = np.array([0, 8, 33, 66, 132, 255], np.uint8)
    tiles = np.zeros((len(vals), 16, 4), np.float32)
    # colours that sit exactly on the 565 grid survive BC1 unchanged
    for i, v in enumerate(vals):
        r5 = (int(v) * 31 + 127) // 255; g6 = (int(v) * 63 + 127) // 255
        tiles[i, :, 0] = r5 / 31.0; tiles[i, :, 1] = g6 / 63.0; tiles[i, :, 2] = r5 / 31.0; tiles[i, :, 3] = 1.0
    dec = oracle.ref_decode_blocks(71, oracle.ref_encode_blocks(71, tiles))
    assert np.allclose(dec, tiles, atol=1e-6)


def test_reference_mip_chain_shapes_and_constant_image():
    _need_ref()
    w, h = 40, 12
    img = np.full((h, w, 4), 77, np.uint8)
    for flt in (0x100000, 0x200000, 0x300000, 0x500000):
        levels = oracle.ref_generate_mips(img, w, h, RGBA8, flt, 6)
        assert [len(x) for x in levels] == [oracle.image_bytes(RGBA8, a, b) for a, b in oracle.mip_sizes(w, h, 6)]
        for lv in levels:
            assert (lv == 77).all(), hex(flt)


def test_reference_box_filter_stale_row_quirk():
    """Generate2DMipsBoxFilter never re-points its fourth tap when the source becomes one texel high
    (DirectXTexMipmaps.cpp:1017 vs :1024-1027): the 1x1 level of a 4x2 image mixes in texel (1,1) of the 4x2 level.
    The HIP path reproduces this; the test pins the behaviour so a change in the oracle build is noticed."""
    _need_ref()
    img = np.zeros((2, 4, 4), np.uint8)
    img[1, 1] = 255                                        # only texel (1,1) is white
    l0, l1, l2 = oracle.ref_generate_mips(img, 4, 2, RGBA8, 0x400000, 3)
    assert l1.reshape(1, 2, 4)[0, 0, 0] == 64              # (0+0+0+255)/4
    assert l2[0] == 96                                     # ((64 + 64) + 0 + 255) / 4 = 95.75: the stale tap, not 32


def test_reference_mse_matches_float64():
    _need_ref()
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (32, 32, 4), dtype=np.uint8); b = rng.integers(0, 256, (32, 32, 4), dtype=np.uint8)
    ref = oracle.ref_compute_mse(a, RGBA8, b, RGBA8, 32, 32)
    f64 = oracle.compute_mse(oracle.load_image(a, 32, 32, RGBA8), oracle.load_image(b, 32, 32, RGBA8))
    assert np.allclose(ref, f64, rtol=1e-4)


def test_reference_error_codes():
    _need_ref()
    img = synth.rgba8(8, 8, seed=1, alpha="opaque")
    with pytest.raises(oracle.RefError) as e:
        oracle.ref_compress_image(img, 8, 8, RGBA8, RGBA8)           # destination not compressed
    assert e.value.hresult == 0x80070057
    with pytest.raises(oracle.RefError) as e:
        oracle.ref_resize(img, 8, 8, RGBA8, 5, 4, 0x400000)          # box needs 2:1
    assert e.value.hresult == 0x80004005


def test_demultiply_alpha_quirk():
    """DemultiplyAlpha (DirectXTexPMAlpha.cpp:134-141): with alpha <= 0 nothing is divided and the select returns the alpha
    splat for RGB - the colour is lost. The GPU kernel reproduces this; pinned here so the oracle keeps saying so."""
    x = np.array([[0.5, 0.25, 0.125, 0.5], [0.3, 0.6, 0.9, 0.0], [0.1, 0.2, 0.3, -1.0]], np.float32)
    r = oracle.ref_premultiply_alpha(x.view(np.uint8).reshape(-1), 3, 1, 2, 2).view(np.float32).reshape(3, 4)
    assert np.array_equal(r, np.array([[1.0, 0.5, 0.25, 0.5], [0, 0, 0, 0], [-1, -1, -1, -1]], np.float32))


def test_alpha_coverage_running_vector_quirk():
    """CalculateAlphaCoverage (DirectXTexMipmaps.cpp:277-289) overwrites the quad's alpha vector with the first sub-sample's sum,
    so a quad (1, 0, 0, 0) - true bilinear coverage at reference 0.5 is 7/64 - feeds sample k+1 with sample k's splatted value.
    A single fully opaque texel among transparent ones must therefore not change the level-1 alpha the way a true bilinear
    estimate would; the test just pins the reference's output on a small asymmetric image."""
    img = np.zeros((4, 4, 4), np.float32); img[..., :3] = 0.5
    img[0, 0, 3] = 1.0; img[1, 2, 3] = 0.8; img[3, 3, 3] = 0.3
    mips = oracle.ref_generate_mips(img, 4, 4, 2, 0x400000, 3)
    out = oracle.ref_scale_mips_alpha_for_coverage(mips, 4, 4, 2, 0.25)
    a1 = out[1].view(np.float32).reshape(2, 2, 4)[..., 3]
    m1 = mips[1].view(np.float32).reshape(2, 2, 4)[..., 3]
    assert np.array_equal(out[0], mips[0])
    scale = a1[0, 0] / m1[0, 0]
    assert np.allclose(a1, m1 * scale, rtol=1e-6) and 0.0 < scale <= 4.0
