"""One image over several contexts of one process (dxtex_compress_multi / dxtex_generate_mips_multi, include/dxtex_amd.h): stripes of block rows
- for the mip filters destination rows with their halo of source rows - on a thread per context must give the bytes of the single-context call
(the in-process form of the split the reference does over OpenMP threads, DirectXTexCompress.cpp:257-281). Three contexts share
GPU 0 here; on a node they would be one per GPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctxs():
    import directxtex_amd as dx
    c = [dx.Context(0) for _ in range(3)]
    yield c
    for x in c:
        x.close()


@pytest.mark.parametrize("fmt,w,h", [(71, 301, 203), (77, 256, 64), (83, 64, 7), (98, 130, 94), (98, 64, 8), (95, 96, 52)])
def test_compress_multi_is_the_single_context_payload(ctxs, fmt, w, h):
    import directxtex_amd as dx
    from directxtex_amd import synth
    if fmt == 95:
        img = synth.survey_rgba16f(w, h, 5); sfmt = 10
    else:
        img = synth.survey_rgba8(w, h, 7, "random"); sfmt = 28
    one = ctxs[0].compress(img, w, h, sfmt, fmt, 0, 0.5)
    for n in (2, 3):
        many = dx.capi.compress_multi(ctxs[:n], img, w, h, sfmt, fmt, 0, 0.5)
        assert np.array_equal(one, many), (fmt, w, h, n)


def test_compress_multi_against_the_reference(ctxs, oracle):
    import directxtex_amd as dx
    from directxtex_amd import synth
    w, h = 200, 120
    img = synth.survey_rgba8(w, h, 11, "opaque")
    got = dx.capi.compress_multi(ctxs, img, w, h, 28, 98, 0, 0.5)
    assert np.array_equal(got, oracle.ref_compress_image(img.reshape(-1), w, h, 28, 98, 0, 0.5))


def test_compress_multi_rejects_bad_calls_before_touching_memory(ctxs):
    """The whole image is validated once, before any stripe pointer is formed (a null image's `pixels + y0 * rowPitch` would pass a stripe's own
    null test and reach the copy); the same context twice is refused; a stripe's failure text reaches the first context."""
    import ctypes
    import directxtex_amd as dx
    from directxtex_amd import capi
    w, h = 64, 64
    rp, sp = capi.compute_pitch(98, w, h)
    out = np.zeros(sp, np.uint8)
    src = capi.Image(w, h, 28, w * 4, w * 4 * h, 0)                       # null pixels
    dst = capi.Image(w, h, 98, rp, sp, out.ctypes.data)
    lib = ctxs[0]._lib
    hr = lib.dxtex_compress_multi(capi._ctx_array(ctxs), 3, ctypes.byref(src), ctypes.byref(dst), 0, ctypes.c_float(0.5))
    assert hr & 0xFFFFFFFF == 0x80004003, hex(hr & 0xFFFFFFFF)         # E_POINTER, not a crash
    assert b"null pixels" in lib.dxtex_ctx_last_error(ctxs[0]._h)
    img = np.zeros((h, w, 4), np.uint8)
    with pytest.raises(dx.DxtexError) as e:
        capi.compress_multi([ctxs[0], ctxs[1], ctxs[0]], img, w, h, 28, 98, 0, 0.5)
    assert e.value.hresult & 0xFFFFFFFF == 0x80070057 and "listed twice" in str(e.value)
    with pytest.raises(dx.DxtexError) as e:
        capi.generate_mips_multi([ctxs[0], ctxs[0]], np.zeros((1024, 64, 4), np.uint8), 64, 1024, 28, 3, 0x400000)
    assert e.value.hresult & 0xFFFFFFFF == 0x80070057


@pytest.mark.parametrize("filt", [0x100000, 0x200000, 0x300000, 0x400000, 0x500000, 0x300000 | 0x2, 0])
@pytest.mark.parametrize("fmt,w,h", [(28, 512, 2048), (28, 1200, 1000), (10, 256, 1024), (2, 64, 1024)])
def test_generate_mips_multi_is_the_single_context_chain(ctxs, filt, fmt, w, h):
    import directxtex_amd as dx
    from directxtex_amd import synth
    pow2 = (w & (w - 1)) == 0 and (h & (h - 1)) == 0
    if (filt & 0xF00000) == 0x400000 and not pow2:
        pytest.skip("the box filter needs power-of-two dimensions")
    if fmt == 28:
        img = synth.survey_rgba8(w, h, 3, "random")
    elif fmt == 10:
        img = synth.survey_rgba16f(w, h, 4)
    else:
        rng = np.random.default_rng(5); img = rng.standard_normal((h, w, 4)).astype(np.float32)
    nlev = 1
    while (w >> nlev) or (h >> nlev):
        nlev += 1
    one = ctxs[0].generate_mips(img, w, h, fmt, nlev, filt)
    many = dx.capi.generate_mips_multi(ctxs, img, w, h, fmt, nlev, filt)
    assert len(one) == len(many)
    for lvl, (a, b) in enumerate(zip(one, many)):
        assert np.array_equal(a, b), (hex(filt), fmt, w, h, lvl)


@pytest.mark.parametrize("filt", [0x300000, 0x400000])
def test_generate_mips_multi_keeps_stripes_resident(ctxs, filt):
    """Round 6: every byte crosses the host link once. A context uploads its rows of level 0 (its stripe plus the halo the split levels
    below need: 3 * 2^k rows k levels up), chains the split levels on the device and downloads its stripes of the levels; nothing is
    re-uploaded per level. Checked through the contexts' own transfer counters."""
    import directxtex_amd as dx
    from directxtex_amd import synth
    w, h, fmt = 1024, 4096, 28
    img = synth.survey_rgba8(w, h, 9, "random")
    nlev = 13
    one = ctxs[0].generate_mips(img, w, h, fmt, nlev, filt)
    for c in ctxs:
        c.transfer_bytes(reset=True)
    many = dx.capi.generate_mips_multi(ctxs, img, w, h, fmt, nlev, filt)
    for lvl, (a, b) in enumerate(zip(one, many)):
        assert np.array_equal(a, b), (hex(filt), lvl)
    moved = [c.transfer_bytes() for c in ctxs]
    n = len(ctxs)
    level0 = w * h * 4
    # split levels: 2048, 1024, 512, 256 rows (>= kSplitMinRows); the tail (128 rows and below) runs on the first context
    split = sum((w >> l) * (h >> l) * 4 for l in range(1, 5))
    tail_up = (w >> 4) * (h >> 4) * 4
    tail_down = sum(max(1, w >> l) * max(1, h >> l) * 4 for l in range(5, nlev))
    halo = 2 * 3 * 16 * w * 4                         # at most 3 * 2^4 rows of level 0 on either side of a stripe
    for i, (up, down) in enumerate(moved):
        extra_up = tail_up if i == 0 else 0
        assert level0 // n <= up - extra_up <= level0 // n + halo + w * 4 * 2, (i, up, level0 // n, halo)
    ups, downs = sum(u for u, _ in moved), sum(d for _, d in moved)
    assert ups <= level0 + n * halo + tail_up + n * w * 8
    assert downs == split + tail_down, (downs, split, tail_down)
