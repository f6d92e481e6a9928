"""The BC7 and BC6H encoders' cores on the host. The BC7 encoder's core (directxtex_amd/csrc/bc7_core.h: seeds, RoughMSE, Refine with PerturbOne / Exhaustive, block
emission, mode choice) compiled for the HOST and run next to the reference's D3DX_BC7::Encode (compiled in place) on random
tiles - flat, gradients, noise of several amplitudes, opaque and with alpha. It checks the shared logic where there is no GPU;
the kernels built on it are checked on the GPU (tests/test_bc7_parity.py)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "bc7_core_check")


@pytest.mark.parametrize("seed,flags", [(7, 0), (11, 0), (12345, 0), (21, 0x80000), (22, 0x80000), (23, 0x100000), (24, 0x180000)])
def test_bc7_core_on_the_host_matches_the_reference(seed, flags):
    """flags: BC_FLAGS (0x80000 BC7_USE_3SUBSETS adds modes 0 and 2, 0x100000 BC7_QUICK leaves mode 6 only)."""
    if not os.path.exists(EXE):
        pytest.fail(f"{EXE} missing: run __graft_entry__.build() where /root/reference exists")
    r = subprocess.run([EXE, "400", str(seed), hex(flags)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 of 400 tiles differ" in r.stdout, r.stdout[-3000:]


LOCKSTEP_EXE = os.path.join(ROOT, "oracle", "_ref", "bc7_lockstep_check")


@pytest.mark.parametrize("seed,flags,flat", [(31, 0, "every"), (32, 0, "every"), (33, 0x80000, "every"), (34, 0x100000, "every"), (35, 0, "kernel"), (36, 0x80000, "kernel")])
def test_bc7_lockstep_pieces_match_the_reference(seed, flags, flat):
    """The same comparison with OptimizeOne taken through the pieces the search kernels run per lane (bc7_core.h: perturb_macro with
    the merged first step, Exhaustive as bound filter + minimum key + exh_advance, the settled-scalar-slot shortcut) instead of the
    straight restatement: what bc7_perturb_kernel / bc7_perturb_filter_kernel / bc7_exhaustive_kernel compute, without a GPU.
    flat = "every": the flat-call shortcut in every mode (checks its argument on all region shapes); "kernel": only where the kernels take it
    (mode 6), so modes 0 - 5 run the kernels' path."""
    if not os.path.exists(LOCKSTEP_EXE):
        pytest.fail(f"{LOCKSTEP_EXE} missing: run __graft_entry__.build() where /root/reference exists")
    r = subprocess.run([LOCKSTEP_EXE, "200", str(seed), hex(flags)], capture_output=True, text=True, timeout=900, env=dict(os.environ, DXTEX_HOST_FLAT_SKIP=flat))
    assert r.returncode == 0 and "0 of 200 tiles differ" in r.stdout, r.stdout[-3000:]


BC6H_EXE = os.path.join(ROOT, "oracle", "_ref", "bc6h_core_check")


@pytest.mark.parametrize("seed", [5, 99])
def test_bc6h_core_on_the_host_matches_the_reference(seed):
    """bc6h_core.h (half -> integer texels, region fits, quantisation per mode, the delta transform and its fit test, PerturbOne,
    block emission) in the order the kernels use it, against D3DX_BC6H::Encode for unsigned and signed targets."""
    if not os.path.exists(BC6H_EXE):
        pytest.fail(f"{BC6H_EXE} missing: run __graft_entry__.build() where /root/reference exists")
    r = subprocess.run([BC6H_EXE, "150", str(seed)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "0 of 300 encodes differ" in r.stdout, r.stdout[-3000:]


BC6H_BOUND_EXE = os.path.join(ROOT, "oracle", "_ref", "bc6h_bound_check")


@pytest.mark.parametrize("seed", [3, 41])
def test_bc6h_perturb_bound_never_exceeds_the_exact_error(seed):
    """perturb6_bound (bc6h_core.h) is what lets bc6h_perturb_filter_kernel skip the exact evaluation of a PerturbOne candidate: it must never be
    above the error MapColorsQuantized's fp32 arithmetic gives, whatever the brightness (tiles from 0.001 to 30000, negatives, flat and noisy,
    unsigned and signed). The host search evaluates both for every candidate (millions) and counts violations; the encodes still match."""
    if not os.path.exists(BC6H_BOUND_EXE):
        pytest.fail(f"{BC6H_BOUND_EXE} missing: run __graft_entry__.build() where /root/reference exists")
    r = subprocess.run([BC6H_BOUND_EXE, "250", str(seed)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "bound above the exact error: 0 candidates" in r.stdout and "0 of 500 encodes differ" in r.stdout, r.stdout[-3000:]
