"""The drop-in claim end to end (runs last): oracle/_ref/binding_demo is the binding INTEGRATION.md documents
(oracle/binding/DirectXTexCompressMI355X.cpp) compiled against the reference's own headers and linked with the reference itself
(libdxtex_ref.so: DirectX::Image, ScratchImage, the CPU DirectX::Compress) and the product (libdxtex_amd.so). The same image goes
through the reference's CPU encoder and through the binding; the two reference ScratchImages must hold identical bytes."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "binding_demo")


def test_reference_types_filled_through_the_binding():
    if not os.path.exists(EXE):
        pytest.fail(f"{EXE} missing: run __graft_entry__.build() where /root/reference exists")
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "binding demo OK" in r.stdout and r.stdout.count("identical ScratchImages") == 8, r.stdout + r.stderr
