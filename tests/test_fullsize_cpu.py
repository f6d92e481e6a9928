"""CPU checks of the full-size golden digests (tests/golden/fullsize.json): the committed file is complete and well-formed, the
generators still produce the inputs it was made from, and - where oracle/_ref exists - the reference reproduces a few of the
digest bands when run on just those rows (a band of 16 block rows is an image of its own for a block codec)."""
import importlib.util
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_golden_fullsize", os.path.join(HERE, "golden", "make_golden_fullsize.py"))
mg = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(mg)
GOLD = json.load(open(os.path.join(HERE, "golden", "fullsize.json")))
CASES = mg.compress_cases()


def test_golden_file_is_complete():
    for cid, kind, w, h, seed, alpha, sfmt, bfmt in CASES:
        g = GOLD["cases"][cid]
        nbh = (h + 3) // 4
        assert g["bytes"] == ((w + 3) // 4) * nbh * 16
        assert len(g["bands"]) == (nbh + mg.BAND_ROWS - 1) // mg.BAND_ROWS
        assert len(g["sha256"]) == 64 and g["ref_seconds"] > 0 and g["ref_threads"] >= 1
    for name in ("box", "cubic"):
        g = GOLD["cases"][f"cfg4_{name}"]
        assert len(g["levels"]) == mg.CFG4["levels"] and len(g["bc3_levels"]) == mg.CFG4["levels"]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_inputs_and_bands(case):
    cid, kind, w, h, seed, alpha, sfmt, bfmt = case
    g = GOLD["cases"][cid]
    img = mg.make_input(kind, w, h, seed, alpha)
    assert mg.sha(img) == g["input_sha256"], cid
    import oracle
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built here")
    rows = mg.BAND_ROWS * 4
    nbands = len(g["bands"])
    for band in (0, nbands // 2 + 1):            # flat-ish and noisy regions both occur in any two bands of these images
        crop = np.ascontiguousarray(img[band * rows:(band + 1) * rows])
        pay = oracle.ref_compress_image(crop, w, crop.shape[0], sfmt, bfmt, mg.TEX_COMPRESS_PARALLEL, 0.5)
        assert mg.sha(pay.reshape(-1, ((w + 3) // 4) * 16)) == g["bands"][band], f"{cid}: band {band}"


def test_huge_goldens_inputs_and_the_seam_band():
    """tests/golden/huge.json (the 16384^2 BC1 image and the two-pass BC7 image): the file is complete, the generator still produces the
    BC7 case's input, and the reference reproduces the digest band that holds the pass seam (block rows 1024 .. 1039: the cut at block 2^22
    falls in row 1024, rows 1024 / 1025 carry the SURVEY recipe) when run on just those rows."""
    _spec = importlib.util.spec_from_file_location("make_golden_huge", os.path.join(HERE, "golden", "make_golden_huge.py"))
    mh = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(mh)
    gold = json.load(open(os.path.join(HERE, "golden", "huge.json")))
    for cid, w, h, seed, noisy, fmt in mh.CASES:
        g = gold["cases"][cid]
        nbw, nbh = (w + 3) // 4, (h + 3) // 4
        assert g["blocks"] == nbw * nbh and g["bytes"] == nbw * nbh * (8 if fmt == mh.BC1 else 16)
        assert len(g["bands"]) == (nbh + mh.BAND_ROWS - 1) // mh.BAND_ROWS and len(g["sha256"]) == 64
    case = [c for c in mh.CASES if c[0] == "huge_bc7_passes"][0]
    cid, w, h, seed, noisy, fmt = case
    g = gold["cases"][cid]
    assert g["blocks"] > (1 << 22) and (1 << 22) // ((w + 3) // 4) == 1024 and (1 << 22) % ((w + 3) // 4) != 0      # the cut is inside block row 1024
    img = mh.make_input(case)
    assert mh.sha(img) == g["input_sha256"]
    import oracle
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built here")
    rows = mh.BAND_ROWS * 4
    band = 1024 // mh.BAND_ROWS
    crop = np.ascontiguousarray(img[band * rows:(band + 1) * rows])
    del img
    pay = oracle.ref_compress_image(crop, w, crop.shape[0], mh.RGBA8, fmt, mh.TEX_COMPRESS_PARALLEL, 0.5)
    assert mh.sha(pay.reshape(-1, ((w + 3) // 4) * 16)) == g["bands"][band]
