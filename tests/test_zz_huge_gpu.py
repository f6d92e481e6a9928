"""The largest shapes the path has to take, at FULL size on the GPU, against the reference's digests (tests/golden/huge.json, produced by
tests/golden/make_golden_huge.py where /root/reference exists; no oracle in the loop):
  * 16384 x 16384 RGBA8 -> BC1: the size limit of DirectXTexImage.cpp:127-131 (1 GiB source, 64 KiB rows, 16 777 216 blocks);
  * 16380 x 4102 RGBA8 -> BC7: 4 201 470 blocks, more than one pass of the search pipeline at the PRODUCT's pass size (2^22 blocks; the
    multi-pass code had only run with DXTEX_MAX_BLOCKS_PER_PASS=17 on a 52 x 36 image before), the cut in the middle of a block row, a
    partial last block row.
The file name sorts late on purpose."""
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_golden_huge", os.path.join(HERE, "golden", "make_golden_huge.py"))
mh = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(mh)
GOLD = json.load(open(os.path.join(HERE, "golden", "huge.json")))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", mh.CASES, ids=[c[0] for c in mh.CASES])
def test_huge_image_identical_to_the_reference(ctx, case):
    cid, w, h, seed, noisy, fmt = case
    gold = GOLD["cases"][cid]
    img = mh.make_input(case)
    assert hashlib.sha256(img.tobytes()).hexdigest() == gold["input_sha256"], "the generator no longer produces the image the digests were made from"
    if fmt == mh.BC7:
        assert gold["blocks"] > (1 << 22)                     # really more than one pass at the product's pass size
    got = ctx.compress(img, w, h, mh.RGBA8, fmt, 0, 0.5)
    del img
    assert got.nbytes == gold["bytes"]
    if hashlib.sha256(got.tobytes()).hexdigest() == gold["sha256"]:
        return
    bands = mh.band_digests(got, w, h, 8 if fmt == mh.BC1 else 16)
    bad = [i for i, (a, b) in enumerate(zip(bands, gold["bands"])) if a != b]
    pytest.fail(f"{cid}: {len(bad)} of {len(bands)} bands of {mh.BAND_ROWS} block rows differ from the reference; first bands {bad[:8]}")
