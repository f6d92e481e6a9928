"""The exact-pruning bounds of the BC7 encoder (bc7_core.h: subset_lower_bound, scalar_kmeans_lower_bound), compiled for
the host: no palette - whatever its endpoints, with the best index per texel - may score below the bound. This checks the
mathematics the pruning rests on; that pruning leaves the encoder's OUTPUT unchanged is tests/test_bc7_parity.py
(test_pruning_changes_nothing, GPU)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "directxtex_amd", "lib", "bc7_bound_check")


def test_lower_bounds_hold_on_random_palettes():
    if not os.path.exists(EXE):
        pytest.fail(f"{EXE} missing: run __graft_entry__.build()")
    r = subprocess.run([EXE, "20000"], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " 0 violations" in r.stdout
    # the same program compares the RGBA end-point fit with its opaque-block variant (bc7_core.h: fit_setup / fit_iterate, A1) bit for bit
    assert " 0 differ between the RGBA fit and its opaque-block variant" in r.stdout


def test_pass_and_segment_construction_invariants():
    """search_common.h: build_passes cuts the block list of an image set into passes and per-image segments, seg_of finds a
    pass-local block's image again. 3 000 random image lists and pass sizes: every block exactly once, in order, lookups right."""
    exe = os.path.join(ROOT, "directxtex_amd", "lib", "passes_check")
    if not os.path.exists(exe):
        pytest.fail(f"{exe} missing: run __graft_entry__.build()")
    r = subprocess.run([exe, "3000"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "blocks checked" in r.stdout, r.stdout[-2000:]


def test_triangle_filter_tables_match_the_reference():
    """The gather lists the host builds for the triangle filter (triangle_filter.h) against the reference's own
    CreateTriangleFilter (filters.h:249-419, compiled in place): same entries, same fp32 weight bits, in the reference's
    accumulation order - for mip-chain steps of odd sizes, arbitrary resizes up and down, 1-texel axes, clamp and wrap."""
    import oracle
    exe = os.path.join(ROOT, "directxtex_amd", "lib", "triangle_check")
    if not os.path.exists(exe):
        pytest.fail(f"{exe} missing: run __graft_entry__.build()")
    rng = np.random.default_rng(31)
    triples = [(s, max(1, s >> 1), w) for s in (1, 2, 3, 5, 7, 9, 17, 33, 100, 127, 255, 1000, 4097) for w in (0, 1)]
    triples += [(int(rng.integers(1, 600)), int(rng.integers(1, 600)), int(rng.integers(0, 2))) for _ in range(300)]
    triples += [(1, 64, 0), (64, 1, 1), (3, 1000, 0), (1000, 3, 1), (2, 2, 0), (8192, 4096, 0)]
    args = [str(v) for t in triples for v in t]
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0
    rows = r.stdout.splitlines()
    assert len(rows) == len(triples)
    entries = 0
    for (source, dest, wrap), row in zip(triples, rows):
        p = row.split()
        ours = [tuple(int(x) for x in e.split(":")) for e in p[1:]]                   # (dst, src, weight bits) in gather order
        s, d, w = oracle.ref_triangle_filter(source, dest, bool(wrap))
        order = np.argsort(d, kind="stable")                                          # per destination, keeping the source-major order
        ref = list(zip(d[order].tolist(), s[order].tolist(), w[order].tolist()))
        assert int(p[0]) == len(ref) and ours == ref, (source, dest, wrap)
        entries += len(ref)
    assert entries > 100000


def test_fit_order_tables_are_size_sorted_permutations():
    """bc67_tables.h: kFit2Order / kFit2Order32 deal the rough passes' fits to wavefronts by subset size (bc7_rough_kernel,
    bc6h_rough_kernel). They must list every subset (shape * 2 + subset) of the two-subset partitions exactly once, largest first -
    a missing or repeated code would leave a fit uncomputed (stale LDS) rather than fail loudly."""
    import re
    text = open(os.path.join(ROOT, "directxtex_amd", "csrc", "bc67_tables.h")).read()

    def table(name):
        body = re.search(name + r"\[\d+\]\s*=\s*\{([^}]*)\}", text).group(1)
        return [int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\d+", body)]

    masks = table("kPart2Mask")
    assert len(masks) == 64

    def size(code):
        ones = bin(masks[code >> 1]).count("1")
        return ones if code & 1 else 16 - ones

    for name, n in (("kFit2Order", 128), ("kFit2Order32", 64)):
        order = table(name)
        assert sorted(order) == list(range(n)), name
        sizes = [size(c) for c in order]
        assert sizes == sorted(sizes, reverse=True), name
        assert min(sizes) >= 3          # no subset of one or two texels among the two-subset shapes: every fit runs OptimizeRGB(A)


def test_folded_cubic_equals_the_plain_one():
    """csrc/cubic_filter.h: cubic_half1 - CUBIC_INTERPOLATE (filters.h:192-207) at dx = 0.5 with the power-of-two products folded into FMAs,
    14 operations instead of 21, what the 2:1 RGBA8 mip kernels evaluate twelve times per texel - returns the bits of the plain form on 20 M
    inputs (8-bit codes / 255, general values, tiny and large magnitudes, flat neighbourhoods)."""
    exe = os.path.join(ROOT, "directxtex_amd", "lib", "cubic_check")
    if not os.path.exists(exe):
        pytest.fail(f"{exe} missing: run __graft_entry__.build()")
    r = subprocess.run([exe, "20000000"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("0 of 20000000 differ"), r.stdout + r.stderr
