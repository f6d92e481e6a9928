"""The drop-in claim end to end (runs last): oracle/_ref/binding_demo is the binding INTEGRATION.md documents
(oracle/binding/DirectXTexCompressMI355X.cpp) compiled against the reference's own headers and linked with the reference itself
(libdxtex_ref.so: DirectX::Image, ScratchImage, the CPU DirectX::Compress) and the product (libdxtex_amd.so). The same image goes
through the reference's CPU encoder and through the binding; the two reference ScratchImages must hold identical bytes.
oracle/binding/DirectXTexMI355X.cpp binds the other entry points of the path the same way - the array Compress (one dxtex_ctx_prepare per
mip size, DirectXTexCompressGPU.cpp:392, then one dxtex_compress_many), Decompress, GenerateMipMaps, Resize, Convert - and the demo
compares each with the reference's own CPU function on reference ScratchImages."""
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
    assert r.returncode == 0 and "binding demo OK" in r.stdout and r.stdout.count("identical ScratchImages") == 16, r.stdout + r.stderr
    # six entry points: Compress (single), Compress (array, with Prepare), Decompress, GenerateMipMaps, Resize, Convert
    for what in ("format 98", "Compress (array 2 x 3 mips) -> BC7", "Decompress BC7", "GenerateMipMaps 44 x 28 triangle", "Resize 44 x 28", "Convert RGBA8"):
        assert any(what in l and "identical ScratchImages" in l for l in r.stdout.splitlines()), (what, r.stdout)
