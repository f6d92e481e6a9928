"""The exact-pruning bounds of the BC7 encoder (bc7_core.h: subset_lower_bound, scalar_kmeans_lower_bound), compiled for
the host: no palette - whatever its endpoints, with the best index per texel - may score below the bound. This checks the
mathematics the pruning rests on; that pruning leaves the encoder's OUTPUT unchanged is tests/test_bc7_parity.py
(test_pruning_changes_nothing, GPU)."""
import os
import subprocess

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


def test_pass_and_segment_construction_invariants():
    """search_common.h: build_passes cuts the block list of an image set into passes and per-image segments, seg_of finds a
    pass-local block's image again. 3 000 random image lists and pass sizes: every block exactly once, in order, lookups right."""
    exe = os.path.join(ROOT, "directxtex_amd", "lib", "passes_check")
    if not os.path.exists(exe):
        pytest.fail(f"{exe} missing: run __graft_entry__.build()")
    r = subprocess.run([exe, "3000"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "blocks checked" in r.stdout, r.stdout[-2000:]
