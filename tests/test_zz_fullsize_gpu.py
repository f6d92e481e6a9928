"""Parity at BASELINE.json's FULL sizes (cfg2, cfg3, cfg4), through the C ABI.

Two kinds of check per configuration:
  * `*_golden`: the HIP output against tests/golden/fullsize.json - SHA-256 digests (whole payload and bands of 16 block rows) of
    the reference's own output, produced by tests/golden/make_golden_fullsize.py where /root/reference exists. No oracle in the
    loop, seconds per case.
  * `*_live`: the reference itself (oracle/_ref: DirectX::Compress with TEX_COMPRESS_PARALLEL -> CompressBC_Parallel,
    DirectXTexCompress.cpp:210-372; GenerateMipMaps, DirectXTexMipmaps.cpp:2828-3017) executed on this box's host cores on the same
    full-size input, compared byte for byte. Minutes per case (the 4096^2 BC7 image is ~3.5 min of reference time on 128 threads);
    DXTEX_SKIP_LIVE_FULLSIZE=1 skips them during development. The reference's wall time goes to gpurun_out/fullsize_live.json.

The file name sorts late on purpose: the long cases run after everything else.
"""
import hashlib
import importlib.util
import json
import os
import time

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_spec = importlib.util.spec_from_file_location("make_golden_fullsize", os.path.join(HERE, "golden", "make_golden_fullsize.py"))
mg = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(mg)
GOLD = json.load(open(os.path.join(HERE, "golden", "fullsize.json")))
CASES = mg.compress_cases()
SKIP_LIVE = os.environ.get("DXTEX_SKIP_LIVE_FULLSIZE") == "1"

pytestmark = pytest.mark.gpu

_inputs = {}


def _input(case):
    cid, kind, w, h, seed, alpha, sfmt, bfmt = case
    if cid not in _inputs:
        _inputs.clear()                                   # one full-size image in memory at a time
        _inputs[cid] = mg.make_input(kind, w, h, seed, alpha)
    return _inputs[cid]


def _record(key, value):
    path = os.path.join(ROOT, "gpurun_out", "fullsize_live.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = value
    json.dump(data, open(path, "w"), indent=1)


def _assert_same_payload(got, ref, w, h, what):
    if np.array_equal(got, ref):
        return
    g = got.reshape(-1, 16); r = ref.reshape(-1, 16)
    bad = np.flatnonzero((g != r).any(axis=1))
    nbw = (w + 3) // 4
    pytest.fail(f"{what}: {bad.size} of {g.shape[0]} blocks differ; first at block ({bad[0] % nbw}, {bad[0] // nbw})")


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_compress_golden(ctx, case):
    cid, kind, w, h, seed, alpha, sfmt, bfmt = case
    g = GOLD["cases"][cid]
    img = _input(case)
    assert mg.sha(img) == g["input_sha256"], "the generator no longer produces the image the golden digests were made from"
    got = ctx.compress(img, w, h, sfmt, bfmt, 0, 0.5)
    assert got.nbytes == g["bytes"]
    bands = mg.band_digests(got, w, h, 16)
    bad = [i for i, (a, b) in enumerate(zip(bands, g["bands"])) if a != b]
    assert not bad, f"{cid}: {len(bad)} of {len(bands)} bands of {mg.BAND_ROWS} block rows differ from the reference; first: band {bad[0]}"
    assert mg.sha(got) == g["sha256"]


@pytest.mark.parametrize("name,flt", [("box", mg.TEX_FILTER_BOX), ("cubic", mg.TEX_FILTER_CUBIC)])
def test_cfg4_chain_golden(ctx, name, flt):
    """8192^2 RGBA8 full chain (14 levels), then BC3 of every level (random alpha)."""
    from directxtex_amd import synth
    c = mg.CFG4
    g = GOLD["cases"][f"cfg4_{name}"]
    if "cfg4" not in _inputs:
        _inputs.clear()
        _inputs["cfg4"] = synth.survey_rgba8(c["width"], c["height"], c["seed"], c["alpha"])
    img = _inputs["cfg4"]
    assert mg.sha(img) == g["input_sha256"]
    levels = ctx.generate_mips(img, c["width"], c["height"], mg.RGBA8, c["levels"], flt)
    assert [mg.sha(l) for l in levels] == g["levels"], f"cfg4 {name} chain differs from the reference"
    w, h = c["width"], c["height"]
    for i, l in enumerate(levels):
        bc3 = ctx.compress(l, w, h, mg.RGBA8, mg.BC3, 0, 0.5)
        assert mg.sha(bc3) == g["bc3_levels"][i], f"cfg4 {name}: BC3 of level {i} ({w}x{h}) differs from the reference"
        w, h = max(1, w >> 1), max(1, h >> 1)


@pytest.mark.skipif(SKIP_LIVE, reason="DXTEX_SKIP_LIVE_FULLSIZE=1")
@pytest.mark.parametrize("case", [c for c in CASES if c[0] != "cfg2_bc7_alpha_2048"], ids=[c[0] for c in CASES if c[0] != "cfg2_bc7_alpha_2048"])
def test_compress_live(ctx, oracle, case):
    """The whole image through the reference's CompressBC_Parallel on this box, byte for byte against the HIP path."""
    cid, kind, w, h, seed, alpha, sfmt, bfmt = case
    img = _input(case)
    got = ctx.compress(img, w, h, sfmt, bfmt, 0, 0.5)
    t0 = time.perf_counter()
    ref = oracle.ref_compress_image(img, w, h, sfmt, bfmt, mg.TEX_COMPRESS_PARALLEL, 0.5)
    dt = time.perf_counter() - t0
    _record(cid, {"ref_seconds": round(dt, 2), "ref_threads": oracle.ref_num_threads(), "Mtexels_s": round(w * h / dt / 1e6, 4),
                  "blocks": int(got.nbytes // 16), "identical": bool(np.array_equal(got, ref))})
    _assert_same_payload(got, ref, w, h, cid)


@pytest.mark.skipif(SKIP_LIVE, reason="DXTEX_SKIP_LIVE_FULLSIZE=1")
def test_cfg4_live(ctx, oracle):
    from directxtex_amd import synth
    c = mg.CFG4
    if "cfg4" not in _inputs:
        _inputs.clear()
        _inputs["cfg4"] = synth.survey_rgba8(c["width"], c["height"], c["seed"], c["alpha"])
    img = _inputs["cfg4"]
    rec = {}
    for name, flt in (("box", mg.TEX_FILTER_BOX), ("cubic", mg.TEX_FILTER_CUBIC)):
        levels = ctx.generate_mips(img, c["width"], c["height"], mg.RGBA8, c["levels"], flt)
        t0 = time.perf_counter()
        ref = oracle.ref_generate_mips(img, c["width"], c["height"], mg.RGBA8, flt, c["levels"])
        rec[f"mips_{name}_seconds"] = round(time.perf_counter() - t0, 2)
        for i, (a, b) in enumerate(zip(levels, ref)):
            assert np.array_equal(a, b), f"cfg4 {name}: level {i} differs from the reference"
        if name == "box":
            w, h = c["width"], c["height"]
            t0 = time.perf_counter()
            for i, l in enumerate(ref):
                rb = oracle.ref_compress_image(l, w, h, mg.RGBA8, mg.BC3, mg.TEX_COMPRESS_PARALLEL, 0.5)
                gb = ctx.compress(levels[i], w, h, mg.RGBA8, mg.BC3, 0, 0.5)
                assert np.array_equal(gb, rb), f"cfg4: BC3 of level {i} ({w}x{h}) differs from the reference"
                w, h = max(1, w >> 1), max(1, h >> 1)
            rec["bc3_chain_seconds_incl_gpu"] = round(time.perf_counter() - t0, 2)
    rec["ref_threads"] = oracle.ref_num_threads()
    _record("cfg4", rec)
