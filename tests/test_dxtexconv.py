"""dxtexconv (directxtex_amd/tools/dxtexconv.cpp): the texconv-style pipeline DDS -> resize -> convert -> mipmaps ->
compress -> DDS on the GPU, compared byte for byte with the same pipeline run through the reference's own functions
(oracle/_ref: Resize, GenerateMipMaps, Compress, SaveToDDSMemory)."""
import os
import subprocess

import numpy as np
import pytest

from directxtex_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "directxtex_amd", "lib", "dxtexconv")
RGBA8 = 28


def _run(args):
    r = subprocess.run([EXE] + args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_resize_mips_bc7(tmp_path, oracle):
    w, h = 100, 60
    img = synth.rgba8(w, h, seed=21, alpha="smooth")
    src = tmp_path / "in.dds"; out = tmp_path / "out.dds"
    oracle.ref_save_dds(img, w, h, RGBA8).tofile(src)
    _run(["-w", "64", "-h", "32", "-m", "0", "-f", "BC7_UNORM", "-if", "CUBIC", "-o", str(out), str(src)])
    small = oracle.ref_resize(img, w, h, RGBA8, 64, 32, 0x300000)
    mips = oracle.ref_generate_mips(small, 64, 32, RGBA8, 0x300000, 7)
    sizes = oracle.mip_sizes(64, 32, 7)
    payload = np.concatenate([oracle.ref_compress_image(m, a, b, RGBA8, 98, 0, 0.5) for m, (a, b) in zip(mips, sizes)])
    want = oracle.ref_save_dds(payload, 64, 32, 98, 1, 7)
    assert np.array_equal(np.fromfile(out, np.uint8), want)


def test_array_bc3_to_bc1_quick_paths(tmp_path, oracle):
    """a 3-item array with mips, block-compressed input. Like texconv (texconv.cpp:2270), -m 0 keeps a chain the input already
    has: every level is decoded and encoded again. Asking for another count (-m 3) rebuilds the chain from the decoded top level."""
    w = h = 32
    imgs = [synth.rgba8(w, h, seed=30 + i, alpha="opaque") for i in range(3)]
    chains = []
    for im in imgs:
        m = oracle.ref_generate_mips(im, w, h, RGBA8, 0x400000, 6)
        chains.append(np.concatenate([oracle.ref_compress_image(x, a, b, RGBA8, 77, 0, 0.5) for x, (a, b) in zip(m, oracle.mip_sizes(w, h, 6))]))
    src = tmp_path / "in.dds"; out = tmp_path / "out.dds"
    oracle.ref_save_dds(np.concatenate(chains), w, h, 77, 3, 6).tofile(src)
    _run(["-m", "0", "-f", "BC1_UNORM", "-if", "BOX", "-o", str(out), str(src)])
    want_chains = []
    for c in chains:
        at = 0
        for (a, b) in oracle.mip_sizes(w, h, 6):
            n = oracle.image_bytes(77, a, b)
            level = oracle.ref_decompress_image(c[at:at + n], a, b, 77, RGBA8)
            want_chains.append(oracle.ref_compress_image(level, a, b, RGBA8, 71, 0, 0.5))
            at += n
    want = oracle.ref_save_dds(np.concatenate(want_chains), w, h, 71, 3, 6)
    assert np.array_equal(np.fromfile(out, np.uint8), want)
    _run(["-m", "3", "-f", "BC1_UNORM", "-if", "BOX", "-y", "-o", str(out), str(src)])
    want_chains = []
    for c in chains:
        top = oracle.ref_decompress_image(c[:oracle.image_bytes(77, w, h)], w, h, 77, RGBA8)
        m = oracle.ref_generate_mips(top, w, h, RGBA8, 0x400000, 3)
        want_chains.append(np.concatenate([oracle.ref_compress_image(x, a, b, RGBA8, 71, 0, 0.5) for x, (a, b) in zip(m, oracle.mip_sizes(w, h, 3))]))
    want = oracle.ref_save_dds(np.concatenate(want_chains), w, h, 71, 3, 3)
    assert np.array_equal(np.fromfile(out, np.uint8), want)


def test_convert_only(tmp_path, oracle):
    w, h = 40, 24
    img = synth.rgba8(w, h, seed=5, alpha="smooth")
    src = tmp_path / "in.dds"; out = tmp_path / "out.dds"
    oracle.ref_save_dds(img, w, h, RGBA8).tofile(src)
    _run(["-f", "R16G16B16A16_FLOAT", "-m", "1", "-o", str(out), str(src)])
    want = oracle.ref_save_dds(oracle.ref_convert(img, w, h, RGBA8, 10), w, h, 10)
    assert np.array_equal(np.fromfile(out, np.uint8), want)


def _bc3_chain(oracle, img, w, h, levels):
    m = oracle.ref_generate_mips(img, w, h, RGBA8, 0x400000, levels)
    return np.concatenate([oracle.ref_compress_image(x, a, b, RGBA8, 77, 0, 0.5) for x, (a, b) in zip(m, oracle.mip_sizes(w, h, levels))])


def test_untouched_blocks_pass_through_and_mips_can_be_trimmed(tmp_path, oracle):
    """texconv keeps the compressed original when no step changes the texels (texconv.cpp:3378-3412, :3566-3574): same format
    and mip count -> the file's blocks come back; -m 1 -> the top level's original blocks, not a re-encode."""
    w = h = 32
    img = synth.rgba8(w, h, seed=41, alpha="smooth")
    chain = _bc3_chain(oracle, img, w, h, 6)
    src = tmp_path / "in.dds"; out = tmp_path / "out.dds"; top = tmp_path / "top.dds"
    oracle.ref_save_dds(chain, w, h, 77, 1, 6).tofile(src)
    _run(["-o", str(out), str(src)])
    assert np.array_equal(np.fromfile(out, np.uint8), np.fromfile(src, np.uint8))
    _run(["-m", "1", "-o", str(top), str(src)])
    assert np.array_equal(np.fromfile(top, np.uint8), oracle.ref_save_dds(chain[:oracle.image_bytes(77, w, h)], w, h, 77, 1, 1))


def test_default_mips_pow2_and_output_directory(tmp_path, oracle):
    """without -m a single-level input gets the full chain (texconv.cpp:2270); -pow2 snaps the size (FitPowerOf2, :1019-1057);
    several inputs go to a directory as <px><name><sx>.dds."""
    w, h = 100, 60
    outdir = tmp_path / "cooked"; outdir.mkdir()
    names = []
    for i in range(2):
        img = synth.rgba8(w, h, seed=50 + i, alpha="opaque")
        src = tmp_path / f"Tex{i}.dds"
        oracle.ref_save_dds(img, w, h, RGBA8).tofile(src)
        names.append((src, img))
    _run(["-pow2", "-px", "p_", "-sx", "_S", "-l", "-y", "-nologo", "-timing", "-o", str(outdir)] + [str(s) for s, _ in names])
    for i, (src, img) in enumerate(names):
        small = oracle.ref_resize(img, w, h, RGBA8, 64, 32, 0)
        mips = oracle.ref_generate_mips(small, 64, 32, RGBA8, 0, 7)
        want = oracle.ref_save_dds(np.concatenate(mips), 64, 32, RGBA8, 1, 7)
        assert np.array_equal(np.fromfile(outdir / f"p_tex{i}_s.dds", np.uint8), want)


def test_keepcoverage_pmalpha_dx10(tmp_path, oracle):
    """mips -> ScaleMipMapsAlphaForCoverage -> PremultiplyAlpha -> BC3, written with the 'DX10' header carrying the alpha mode."""
    w = h = 64
    img = synth.rgba8(w, h, seed=61, alpha="smooth")
    src = tmp_path / "in.dds"; out = tmp_path / "out.dds"
    oracle.ref_save_dds(img, w, h, RGBA8).tofile(src)
    _run(["-m", "0", "-if", "BOX", "-keepcoverage", "0.5", "-pmalpha", "-f", "BC3_UNORM", "-dx10", "-o", str(out), str(src)])
    sizes = oracle.mip_sizes(w, h, 7)
    mips = oracle.ref_generate_mips(img, w, h, RGBA8, 0x400000, 7)
    cov = oracle.ref_scale_mips_alpha_for_coverage(mips, w, h, RGBA8, 0.5)
    pm = [oracle.ref_premultiply_alpha(m, a, b, RGBA8, 0) for m, (a, b) in zip(cov, sizes)]
    bc = np.concatenate([oracle.ref_compress_image(m, a, b, RGBA8, 77, 0, 0.5) for m, (a, b) in zip(pm, sizes)])
    hr, want = oracle.ref_save_dds_ex(bc, w, h, 1, 77, 1, 7, 0, 2, 3, 0x10000 | 0x20000)            # alpha mode 2 = premultiplied
    assert hr == 0
    assert np.array_equal(np.fromfile(out, np.uint8), want)
    # an opaque texture is tagged opaque (alpha mode 3), whatever was asked for
    img2 = synth.rgba8(w, h, seed=62, alpha="opaque")
    oracle.ref_save_dds(img2, w, h, RGBA8).tofile(src)
    _run(["-m", "1", "-dx10", "-y", "-o", str(out), str(src)])
    hr, want = oracle.ref_save_dds_ex(img2, w, h, 1, RGBA8, 1, 1, 0, 3, 3, 0x10000 | 0x20000)
    assert np.array_equal(np.fromfile(out, np.uint8), want)


def test_hdr_to_bc6h(tmp_path, oracle):
    """a Radiance .hdr file (the reference's own writer) -> full mip chain -> BC6H_UF16, and back out as .hdr from a float DDS."""
    w, h = 64, 32
    rng = np.random.default_rng(71)
    img = (rng.random((h, w, 4), dtype=np.float32) * np.exp2(rng.integers(-3, 5, (h, w, 1))).astype(np.float32)).astype(np.float32)
    hr, hdr = oracle.ref_save_hdr(img, w, h, 2, w * 16)
    assert hr == 0
    src = tmp_path / "sky.hdr"; out = tmp_path / "sky.dds"
    hdr.tofile(src)
    _run(["-f", "BC6H_UF16", "-if", "BOX", "-o", str(out), str(src)])
    hr, meta, px = oracle.ref_load_hdr(hdr)
    mips = oracle.ref_generate_mips(px, w, h, 2, 0x400000, 7)
    payload = np.concatenate([oracle.ref_compress_image(m, a, b, 2, 95, 0, 0.5) for m, (a, b) in zip(mips, oracle.mip_sizes(w, h, 7))])
    want = oracle.ref_save_dds(payload, w, h, 95, 1, 7)
    assert np.array_equal(np.fromfile(out, np.uint8), want)
    # float DDS -> .hdr
    f32 = tmp_path / "f32.dds"; back = tmp_path / "back.hdr"
    oracle.ref_save_dds(px, w, h, 2).tofile(f32)
    _run(["-m", "1", "-ft", "hdr", "-o", str(back), str(f32)])
    assert np.array_equal(np.fromfile(back, np.uint8), oracle.ref_save_hdr(px, w, h, 2, w * 16)[1])


def test_tga_to_bc7_and_back_to_tga(tmp_path, oracle):
    """a 32-bit run-length-free TGA (the reference's writer) -> BC7 with mips; an RGBA8 DDS -> TGA 2.0 with the alpha mode set."""
    w, h = 48, 32
    img = synth.rgba8(w, h, seed=81, alpha="smooth")
    hr, tga = oracle.ref_save_tga(img, w, h, RGBA8, w * 4)
    assert hr == 0
    src = tmp_path / "albedo.tga"; out = tmp_path / "albedo.dds"
    tga.tofile(src)
    _run(["-f", "BC7_UNORM", "-if", "CUBIC", "-m", "3", "-o", str(out), str(src)])
    hr, meta, px = oracle.ref_load_tga(tga)
    assert np.array_equal(px, img.reshape(-1))
    mips = oracle.ref_generate_mips(px, w, h, RGBA8, 0x300000, 3)
    payload = np.concatenate([oracle.ref_compress_image(m, a, b, RGBA8, 98, 0, 0.5) for m, (a, b) in zip(mips, oracle.mip_sizes(w, h, 3))])
    assert np.array_equal(np.fromfile(out, np.uint8), oracle.ref_save_dds(payload, w, h, 98, 1, 3))
    dds = tmp_path / "rgba.dds"; back = tmp_path / "back.tga"
    oracle.ref_save_dds(img, w, h, RGBA8).tofile(dds)
    _run(["-m", "1", "-ft", "tga", "-o", str(back), str(dds)])
    assert np.array_equal(np.fromfile(back, np.uint8), oracle.ref_save_tga(img, w, h, RGBA8, w * 4)[1])          # plain TGA: no extension area
    _run(["-m", "1", "-ft", "tga", "-tga20", "-y", "-o", str(back), str(dds)])
    ours = np.fromfile(back, np.uint8)
    hr, want = oracle.ref_save_tga(img, w, h, RGBA8, w * 4, 0, 1)               # alpha mode straight, as texconv tags it
    at = want.size - 26 - 495
    ours[at + 367:at + 379] = 0; want = want.copy(); want[at + 367:at + 379] = 0          # the time stamp
    assert np.array_equal(ours, want)


def test_files_dealt_out_over_worker_contexts(tmp_path, oracle):
    """-gpus a,b,...: one host thread and one context per entry, file i on worker i mod n, nothing shared. On a
    one-GPU box the two workers are two contexts on the same device - the host-side concurrency is what is tested here."""
    w = h = 64
    outdir = tmp_path / "out"; outdir.mkdir()
    srcs = []
    for i in range(5):
        img = synth.rgba8(w, h, seed=100 + i, alpha="smooth")
        p = tmp_path / f"t{i}.dds"
        oracle.ref_save_dds(img, w, h, RGBA8).tofile(p)
        srcs.append((p, img))
    _run(["-gpus", "0,0", "-f", "BC7_UNORM", "-m", "1", "-nologo", "-o", str(outdir)] + [str(p) for p, _ in srcs])
    for i, (p, img) in enumerate(srcs):
        want = oracle.ref_save_dds(oracle.ref_compress_image(img, w, h, RGBA8, 98, 0, 0.5), w, h, 98)
        assert np.array_equal(np.fromfile(outdir / f"t{i}.dds", np.uint8), want), i


def test_the_chain_stays_on_the_device(tmp_path, oracle):
    """resize -> mipmaps -> keepcoverage -> BC3 on ONE uploaded copy (DeviceScratchImage): the bytes are those of the
    reference's step-by-step pipeline, and what crosses PCIe is the decoded source once and the final payload once (+ 8-byte
    reduction results: the alpha scans and the coverage bisection) - `-timing` prints the context's transfer counters."""
    import re
    w, h = 200, 120
    img = synth.rgba8(w, h, seed=131, alpha="smooth")
    src = tmp_path / "in.dds"; out = tmp_path / "out.dds"
    oracle.ref_save_dds(img, w, h, RGBA8).tofile(src)
    txt = _run(["-w", "128", "-h", "64", "-m", "0", "-if", "CUBIC", "-keepcoverage", "0.4", "-f", "BC3_UNORM", "-timing", "-overlap", "1", "-o", str(out), str(src)])
    small = oracle.ref_resize(img, w, h, RGBA8, 128, 64, 0x300000)
    sizes = oracle.mip_sizes(128, 64, 8)
    mips = oracle.ref_generate_mips(small, 128, 64, RGBA8, 0x300000, 8)
    cov = oracle.ref_scale_mips_alpha_for_coverage(mips, 128, 64, RGBA8, 0.4)
    payload = np.concatenate([oracle.ref_compress_image(m, a, b, RGBA8, 77, 0, 0.5) for m, (a, b) in zip(cov, sizes)])
    assert np.array_equal(np.fromfile(out, np.uint8), oracle.ref_save_dds(payload, 128, 64, 77, 1, 8))
    m = re.search(r"host -> device (\d+) bytes, device -> host (\d+) bytes", txt)
    assert m, txt
    up, down = int(m.group(1)), int(m.group(2))
    assert up == w * h * 4, (up, w * h * 4)
    assert payload.size <= down <= payload.size + 8 * (2 + 1 + 10 * 7), (down, payload.size)


def test_two_workers_per_gpu_by_default(tmp_path, oracle):
    """more files than GPUs: two contexts per GPU take turns, so one file's codec work and transfers overlap another's kernels; the
    outputs are those of one worker."""
    w = h = 48
    outdir = tmp_path / "out"; outdir.mkdir()
    srcs = []
    for i in range(4):
        img = synth.rgba8(w, h, seed=140 + i, alpha="opaque")
        p = tmp_path / f"u{i}.dds"
        oracle.ref_save_dds(img, w, h, RGBA8).tofile(p)
        srcs.append((p, img))
    _run(["-f", "BC1_UNORM", "-m", "1", "-nologo", "-o", str(outdir)] + [str(p) for p, _ in srcs])
    for i, (p, img) in enumerate(srcs):
        want = oracle.ref_save_dds(oracle.ref_compress_image(img, w, h, RGBA8, 71, 0, 0.5), w, h, 71)
        assert np.array_equal(np.fromfile(outdir / f"u{i}.dds", np.uint8), want), i
