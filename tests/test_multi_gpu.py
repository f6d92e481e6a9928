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
