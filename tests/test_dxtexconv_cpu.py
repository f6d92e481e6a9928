"""dxtexconv without a GPU: the header query (-info), option parsing, and the refusal to run any image step without a
gfx950 device (the tool has no CPU path)."""
import os
import subprocess

import numpy as np

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "directxtex_amd", "lib", "dxtexconv")


def run(args):
    return subprocess.run([EXE] + args, capture_output=True, text=True, timeout=60)


def test_info_reads_headers_of_all_three_containers(tmp_path):
    rng = np.random.default_rng(1)
    d = str(tmp_path)
    oracle.ref_save_dds(rng.integers(0, 256, oracle.texture_bytes(77, 20, 12, 6, 3), dtype=np.uint8), 20, 12, 77, 6, 3, 4).tofile(d + "/cube.dds")
    oracle.ref_save_hdr(rng.random((4, 16, 4), dtype=np.float32), 16, 4, 2, 256)[1].tofile(d + "/sky.hdr")
    oracle.ref_save_tga(rng.integers(0, 256, (3, 5, 4), dtype=np.uint8), 5, 3, 28, 20, 0x20, 2)[1].tofile(d + "/a.tga")
    oracle.ref_save_dds_ex(rng.integers(0, 256, 8 * 4 * 2 * 2 + 4 * 2 * 2, dtype=np.uint8), 8, 4, 2, 53, 1, 2, 0, 0, 4, 0)[1].tofile(d + "/vol565.dds")      # R16_TYPELESS: held by the container, no kernels
    with open(d + "/bad.dds", "wb") as f:
        f.write(b"nope")
    r = run(["-info", d + "/cube.dds", d + "/sky.hdr", d + "/a.tga", d + "/vol565.dds", d + "/bad.dds"])
    out = r.stdout.replace(d, ".").splitlines()
    assert r.returncode == 1                      # one file could not be read
    assert out[0] == "./cube.dds: 20x12 cube mips 3 items 6 format 77 BC3_UNORM bpp 8 alpha unknown images 18 bytes 2208"
    assert out[1] == "./sky.hdr: 16x4 2D mips 1 items 1 format 2 R32G32B32A32_FLOAT bpp 128 alpha opaque images 1 bytes 1024"
    assert out[2] == "./a.tga: 5x3 2D mips 1 items 1 format 29 R8G8B8A8_UNORM_SRGB bpp 32 alpha premultiplied sRGB images 1 bytes 60"
    assert out[3].startswith("./vol565.dds: 8x4x2 3D mips 2 items 1 format 53") and out[3].endswith("(container only: no GPU path for this format)")
    assert out[4] == "./bad.dds: FAILED (80004005)"
    assert run(["-info", d + "/cube.dds"]).returncode == 0


def test_option_errors():
    for args in (["-o", "x.dds"], ["in.dds"], ["-f", "NOT_A_FORMAT", "-o", "x.dds", "in.dds"], ["-if", "CUBIC_DITHER", "-o", "x.dds", "in.dds"],
                 ["-wrap", "-mirror", "-o", "x.dds", "in.dds"], ["-pmalpha", "-alpha", "-o", "x.dds", "in.dds"], ["-dx10", "-dx9", "-o", "x.dds", "in.dds"],
                 ["-gpus", "0,x", "-o", "x.dds", "in.dds"], ["-fl", "13.0", "-o", "x.dds", "in.dds"], ["-keepcoverage", "2", "-o", "x.dds", "in.dds"],
                 ["-hflip", "-o", "x.dds", "in.dds"], ["-ft", "png", "-o", "x.dds", "in.dds"], ["-w"], ["-bc", "z", "-o", "x.dds", "in.dds"]):
        r = run(args)
        assert r.returncode == 1 and "usage: dxtexconv" in r.stderr, (args, r.stderr)
