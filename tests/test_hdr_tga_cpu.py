"""The Radiance .hdr codec of the C++ host layer, differentially against the reference's DirectXTexHDR.cpp (compiled in place
into oracle/_ref): files written are byte-identical, files read give bit-identical floats, and malformed, truncated and
mutated files produce the same HRESULT. CPU only."""
import os
import subprocess

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "directxtex_amd", "lib", "host_api_test")
RGBA32F, RGB32F, RGBA16F = 2, 6, 10


def load_many(tmp, kind, files, flags=0):
    lines = []
    for i, data in enumerate(files):
        path = os.path.join(tmp, f"f{i}.{kind}")
        with open(path, "wb") as f:
            f.write(bytes(data))
        lines.append(f"{path} {flags}")
    lst = os.path.join(tmp, "list.txt")
    with open(lst, "w") as f:
        f.write("\n".join(lines) + "\n")
    r = subprocess.run([EXE, "codec_load_many", kind, lst], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = r.stdout.splitlines()
    assert len(rows) == len(files)
    out = []
    for i, row in enumerate(rows):
        p = row.split()
        if len(p) > 2:
            out.append((int(p[1], 16), dict(zip(("width", "height", "format", "miscFlags2"), (int(x) for x in p[3:]))), np.fromfile(os.path.join(tmp, f"f{i}.{kind}.out"), np.uint8)))
        else:
            out.append((int(p[1], 16), None, None))
    return out


def save(tmp, kind, px, w, h, fmt, row_pitch, flags=0):
    src = os.path.join(tmp, "px.bin"); out = os.path.join(tmp, f"out.{kind}")
    np.ascontiguousarray(px).view(np.uint8).reshape(-1).tofile(src)
    if os.path.exists(out):
        os.remove(out)
    r = subprocess.run([EXE, "codec_save", kind, src, str(w), str(h), str(fmt), str(row_pitch), str(flags), out], capture_output=True, text=True, timeout=60)
    hr = int(r.stdout.split()[1], 16)
    return hr, (np.fromfile(out, np.uint8) if hr == 0 else None)


def hdr_images(rng):
    """float images that exercise the encoder: smooth ramps (runs), noise (literals), flat areas, negatives, zeros, huge and tiny values."""
    for (w, h) in ((16, 4), (8, 3), (7, 5), (300, 2), (129, 3), (1, 1), (64, 1)):
        noise = rng.random((h, w, 4), dtype=np.float32) * np.float32(4.0)
        ramp = np.broadcast_to((np.arange(w, dtype=np.float32) // 5)[None, :, None] * np.float32(0.125), (h, w, 4)).copy()
        flat = np.full((h, w, 4), 0.5, np.float32)
        mixed = np.where(rng.random((h, w, 1)) < 0.5, ramp, noise).astype(np.float32)
        mixed[..., 1] = np.float32(0.25)
        wild = (rng.standard_normal((h, w, 4)) * np.exp2(rng.integers(-40, 40, (h, w, 1)))).astype(np.float32)
        wild[rng.random((h, w)) < 0.2] = 0
        yield w, h, (noise, ramp, flat, mixed, wild)


def test_hdr_writer_is_byte_identical_and_files_round_trip(tmp_path):
    rng = np.random.default_rng(11)
    n = 0
    for w, h, imgs in hdr_images(rng):
        for img in imgs:
            for fmt in (RGBA32F, RGB32F, RGBA16F):
                if fmt == RGBA32F:
                    px = img; pitch = w * 16
                elif fmt == RGB32F:
                    px = np.ascontiguousarray(img[..., :3]); pitch = w * 12
                else:
                    with np.errstate(over="ignore"):
                        px = np.clip(img, -60000, 60000).astype(np.float16); pitch = w * 8
                hr, ours = save(str(tmp_path), "hdr", px, w, h, fmt, pitch)
                rhr, ref = oracle.ref_save_hdr(px, w, h, fmt, pitch)
                assert hr == rhr == 0
                assert np.array_equal(ours, ref), (w, h, fmt)
                n += 1
    assert n > 60
    # padded rows, and what cannot be saved
    img = rng.random((3, 9, 4), dtype=np.float32)
    padded = np.zeros((3, 200), np.uint8); padded[:, :144] = img.view(np.uint8).reshape(3, 144)
    hr, ours = save(str(tmp_path), "hdr", padded, 9, 3, RGBA32F, 200)
    assert hr == 0 and np.array_equal(ours, oracle.ref_save_hdr(img, 9, 3, RGBA32F, 144)[1])
    assert save(str(tmp_path), "hdr", np.zeros((2, 2, 4), np.uint8), 2, 2, 28, 8)[0] == oracle.ref_save_hdr(np.zeros((2, 2, 4), np.uint8), 2, 2, 28, 8)[0] == 0x80070032


def _rows_old_rle(rng, w, h):
    """scanlines in the old scheme: texels and (1,1,1,n) repeat markers; a marker that directly follows another counts n << 8."""
    out = bytearray()
    for _ in range(h):
        x = 0
        after_marker = True            # a row starts with a texel
        while x < w:
            left = w - x
            if not after_marker and left >= 2 and rng.random() < 0.4:
                n = int(rng.integers(1, min(left, 255) + 1))
                out += bytes([1, 1, 1, n]); x += n
                if w - x >= 256 and rng.random() < 0.5:
                    out += bytes([1, 1, 1, 1]); x += 256           # chained: 1 << 8
                after_marker = True
            else:
                out += bytes(int(v) for v in rng.integers(3, 256, 4))
                x += 1
                after_marker = False
    return bytes(out)


def test_hdr_reader_matches_the_reference(tmp_path):
    rng = np.random.default_rng(12)
    files = []
    # files of the reference's own writer (new run-length scheme and raw rows)
    for w, h, imgs in hdr_images(rng):
        for img in imgs:
            files.append(oracle.ref_save_hdr(img, w, h, RGBA32F, w * 16)[1].tobytes())
    # header grammar
    body = lambda w, h: bytes(int(v) for v in rng.integers(3, 256, w * h * 4))
    heads = [b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 3 +X 5\n", b"#?RGBE\nFORMAT=32-bit_rle_xyze\n\n-Y 3 +X 5\n",
             b"#?RADIANCE\n# a comment\nSOFTWARE=x\nEXPOSURE=2.5\nFORMAT= \t32-bit_rle_rgbe\nEXPOSURE= 0.5\nEXPOSURE=1e20\nEXPOSURE=abc\n\n-Y 3 +X 5\n",
             b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n+Y 3 +X 5\n", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 3 -X 5\n", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 3 +Y 5\n",
             b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-X 3 +Y 5\n", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\nQQ 3 +X 5\n", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y +X 5\n",
             b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 3 +X\n", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 3\n", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 0 +X 5\n",
             b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 70000 +X 5\n", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 3 +X 70000\n", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y -3 +X 5\n",
             b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 40000 +X 40000\n", b"#?RADIANCE\nFORMAT=32-bit_rgbe\n\n-Y 3 +X 5\n", b"#?RADIANCE\n\n-Y 3 +X 5\n",
             b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n-Y 3 +X 5\n", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 3 +X 5", b"#?RADIANC\n",
             b"#?RADIANCE\nFORMAT=\n\n-Y 3 +X 5\n", b"#?RADIANCE\nEXPOSURE=\nFORMAT=32-bit_rle_rgbe\n\n-Y 3 +X 5\n", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\x00\n-Y 3 +X 5\n",
             b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y   3   +X   5   \n", b"#?RADIANCE\r\nFORMAT=32-bit_rle_rgbe\r\n\n-Y 3 +X 5\n", b"#?RADIANCEFORMAT=32-bit_rle_rgbe\n\n-Y 3 +X 5\n"]
    for hd in heads:
        files.append(hd + body(5, 3))
        files.append(hd)
    # old-scheme run lengths
    for (w, h) in ((20, 3), (600, 2), (9, 4)):
        for _ in range(6):
            files.append(f"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y {h} +X {w}\n".encode() + _rows_old_rle(rng, w, h))
    # truncations of good files
    good = oracle.ref_save_hdr(next(iter(hdr_images(np.random.default_rng(3))))[2][3], 16, 4, RGBA32F, 256)[1].tobytes()
    for cut in list(range(0, 60)) + list(range(60, len(good), 7)):
        files.append(good[:cut])
    ours = load_many(str(tmp_path), "hdr", files)
    loaded = 0
    for i, (data, (hr, meta, px)) in enumerate(zip(files, ours)):
        rhr, rmeta, rpx = oracle.ref_load_hdr(data)
        assert hr == rhr, (i, hex(hr), hex(rhr), data[:80])
        assert meta == rmeta, (i, meta, rmeta)
        if rpx is not None:
            assert np.array_equal(px, rpx), (i, data[:80])
            loaded += 1
    assert loaded > 55


def test_hdr_seeded_mutations(tmp_path):
    rng = np.random.default_rng(13)
    bases = []
    for w, h, imgs in hdr_images(np.random.default_rng(4)):
        bases.append(oracle.ref_save_hdr(imgs[3], w, h, RGBA32F, w * 16)[1].tobytes())
        bases.append(f"#?RADIANCE\nEXPOSURE=3\nFORMAT=32-bit_rle_rgbe\n\n-Y {h} +X {w}\n".encode() + _rows_old_rle(np.random.default_rng(w), w, h))
    files = []
    for i in range(2500):
        b = bytearray(bases[i % len(bases)])
        for _ in range(int(rng.integers(1, 4))):
            at = int(rng.integers(0, min(len(b), 120))) if rng.random() < 0.6 else int(rng.integers(0, len(b)))
            b[at] = int(rng.choice([0, 1, 2, 10, 32, 43, 45, 48, 57, 88, 89, 127, 128, 129, 255, int(rng.integers(0, 256))]))
        if rng.random() < 0.2:
            del b[int(rng.integers(0, len(b))):]
        files.append(bytes(b))
    ours = load_many(str(tmp_path), "hdr", files)
    for i, (data, (hr, meta, px)) in enumerate(zip(files, ours)):
        rhr, rmeta, rpx = oracle.ref_load_hdr(data)
        assert (hr, meta) == (rhr, rmeta), (i, hex(hr), hex(rhr), data[:100])
        if rpx is not None:
            assert np.array_equal(px, rpx), (i, data[:100])


# ---- TGA -----------------------------------------------------------------------------------------------------------------------
import struct

TGA_STAMP = slice(367, 379)          # the time stamp inside the TGA 2.0 extension area: the only bytes that may differ


def tga_load_many(tmp, files, flags):
    lines = []
    for i, data in enumerate(files):
        path = os.path.join(tmp, f"t{i}.tga")
        with open(path, "wb") as f:
            f.write(bytes(data))
        lines.append(f"{path} {flags}")
    lst = os.path.join(tmp, "list.txt")
    with open(lst, "w") as f:
        f.write("\n".join(lines) + "\n")
    r = subprocess.run([EXE, "codec_load_many", "tga", lst], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = r.stdout.splitlines()
    assert len(rows) == len(files)
    out = []
    for i, row in enumerate(rows):
        p = row.split()
        if len(p) > 2:
            out.append((int(p[1], 16), dict(zip(oracle.TGA_META_KEYS, (int(x) for x in p[3:]))), np.fromfile(os.path.join(tmp, f"t{i}.tga.out"), np.uint8)))
        else:
            out.append((int(p[1], 16), None, None))
    return out


def tga_compare(tmp, files, flags, what):
    ours = tga_load_many(tmp, files, flags)
    loaded = 0
    for i, (data, (hr, meta, px)) in enumerate(zip(files, ours)):
        rhr, rmeta, rpx = oracle.ref_load_tga(data, flags)
        where = (what, i, hex(flags), bytes(data[:18]).hex())
        assert hr == rhr, (hex(hr), hex(rhr), where)
        assert meta == rmeta, (meta, rmeta, where)
        if rpx is not None:
            assert np.array_equal(px, rpx), where
            loaded += 1
    return loaded


def tga_header(image_type, w, h, bpp, descriptor=0, id_len=0, cmap_type=0, cmap_first=0, cmap_len=0, cmap_size=0):
    return struct.pack("<BBBHHBHHHHBB", id_len, cmap_type, image_type, cmap_first, cmap_len, cmap_size, 0, 0, w, h, bpp, descriptor)


def tga_rle(rng, pixels, bpp_bytes, width):
    """pixels: (n, bpp_bytes) uint8 rows-major; packets never cross rows."""
    out = bytearray()
    n = pixels.shape[0]
    for row in range(n // width):
        x = 0
        while x < width:
            left = width - x
            k = int(rng.integers(1, min(left, 128) + 1))
            at = row * width + x
            if rng.random() < 0.5:
                out += bytes([0x80 | (k - 1)]) + pixels[at].tobytes()
                pixels[at:at + k] = pixels[at]
            else:
                out += bytes([k - 1]) + pixels[at:at + k].tobytes()
            x += k
    return bytes(out)


def tga_extension(attributes, gamma=(0, 0), size=495):
    ext = bytearray(495)
    struct.pack_into("<H", ext, 0, size)
    struct.pack_into("<HH", ext, 478, *gamma)
    ext[494] = attributes
    return bytes(ext)


def tga_footer(ext_offset, signature=b"TRUEVISION-XFILE.\0"):
    return struct.pack("<II", ext_offset, 0) + signature


def tga_files(rng):
    """hand-built files: every accepted type and depth, raw and run-length encoded, all four orientations, ID fields, palettes,
    alpha patterns (all zero, all opaque, mixed), extension areas; and the headers that must be refused."""
    files = []
    for (w, h) in ((7, 5), (1, 1), (130, 3)):
        for desc_bits in (0, 0x10, 0x20, 0x30):
            for (itype, bpp) in ((2, 32), (2, 24), (2, 16), (3, 8), (10, 32), (10, 24), (10, 16), (11, 8)):
                B = bpp // 8
                px = rng.integers(0, 256, (w * h, B), dtype=np.uint8)
                pattern = int(rng.integers(0, 3))
                if B == 4:
                    px[:, 3] = (0, 255, px[:, 3])[pattern] if pattern < 2 else px[:, 3]
                if B == 2:
                    px[:, 1] = (px[:, 1] & 0x7F, px[:, 1] | 0x80, px[:, 1])[pattern]
                body = tga_rle(rng, px, B, w) if itype >= 9 else px.tobytes()
                idf = bytes(int(v) for v in rng.integers(0, 256, 5)) if rng.random() < 0.3 else b""
                f = tga_header(itype, w, h, bpp, desc_bits | (8 if bpp == 32 else 0), id_len=len(idf)) + idf + body
                files.append(f)
                # with a TGA 2.0 footer + extension area
                for attr, gamma, size in ((3, (22, 10), 495), (4, (24, 10), 495), (1, (1, 1), 495), (2, (0, 0), 495), (0, (22, 10), 494), (7, (219, 100), 495)):
                    if rng.random() < 0.25:
                        files.append(f + tga_extension(attr, gamma, size) + tga_footer(len(f)))
                if rng.random() < 0.2:
                    files.append(f + tga_extension(3, (22, 10)) + tga_footer(len(f) + 100))          # offset past the end
                    files.append(f + tga_extension(3, (22, 10)) + tga_footer(len(f), b"TRUEVISION-XFILE-\0"))
                    files.append(f + tga_footer(0))
        # colour-mapped
        for (first, length) in ((0, 256), (0, 16), (10, 20), (200, 56), (200, 57), (0, 257)):
            idx = rng.integers(0, 256, w * h, dtype=np.uint8)
            cmap = rng.integers(0, 256, length * 3, dtype=np.uint8).tobytes()
            files.append(tga_header(1, w, h, 8, 0x20, cmap_type=1, cmap_first=first, cmap_len=length, cmap_size=24) + cmap + idx.tobytes())
            files.append(tga_header(1, w, h, 8, 0x00, id_len=3, cmap_type=1, cmap_first=first, cmap_len=length, cmap_size=24) + b"abc" + cmap + idx.tobytes())
            files.append(tga_header(1, w, h, 8, 0x20, cmap_type=1, cmap_first=first, cmap_len=length, cmap_size=24) + cmap[:len(cmap) // 2])
    # refused headers
    body = bytes(4096)
    for args in ((0, 4, 4, 32), (9, 4, 4, 8), (2, 4, 4, 8), (2, 4, 4, 15), (3, 4, 4, 16), (2, 0, 4, 32), (2, 4, 0, 32), (4, 4, 4, 32), (12, 4, 4, 32), (255, 4, 4, 32)):
        files.append(tga_header(*args) + body)
    files.append(tga_header(2, 4, 4, 32, 0x40) + body); files.append(tga_header(2, 4, 4, 32, 0x80) + body)
    files.append(tga_header(2, 4, 4, 32, cmap_type=1) + body); files.append(tga_header(2, 4, 4, 32, cmap_len=5) + body)
    files.append(tga_header(1, 4, 4, 8, cmap_type=1, cmap_len=16, cmap_size=32) + body); files.append(tga_header(1, 4, 4, 8, cmap_type=0, cmap_len=16, cmap_size=24) + body)
    files.append(tga_header(1, 4, 4, 16, cmap_type=1, cmap_len=16, cmap_size=24) + body); files.append(tga_header(3, 4, 4, 8, cmap_len=1) + body)
    files.append(tga_header(2, 65535, 65535, 32) + body); files.append(tga_header(2, 4, 4, 32, id_len=255) + body[:100]); files.append(tga_header(2, 4, 4, 32))
    files.append(tga_header(2, 4, 4, 32)[:17]); files.append(b"\0")
    return files


@pytest.mark.parametrize("flags", [0, 0x1, 0x2, 0x10, 0x80, 0x1 | 0x2 | 0x80])
def test_tga_reader_matches_the_reference(tmp_path, flags):
    rng = np.random.default_rng(21)
    files = tga_files(rng)
    # truncations of a run-length encoded file and of a raw one
    for base in (files[4], files[0], files[6]):
        for cut in sorted(set(list(range(0, 30)) + list(range(30, len(base), 5)) + [len(base) - 1])):
            files.append(base[:cut])
    loaded = tga_compare(str(tmp_path), files, flags, "built")
    assert loaded > 150


def test_tga_writer_is_byte_identical(tmp_path):
    rng = np.random.default_rng(22)
    n = 0
    for (w, h) in ((5, 3), (1, 1), (64, 2)):
        for fmt, B in ((28, 4), (29, 4), (87, 4), (91, 4), (88, 4), (93, 4), (61, 1), (65, 1), (86, 2), (2, 16), (49, 2)):
            px = rng.integers(0, 256, (h, w * B), dtype=np.uint8)
            for flags, alpha_mode in ((0, -1), (0, 0), (0, 1), (0, 2), (0, 3), (0, 4), (0x20, 1), (0x40, 0), (0x20, -1), (0x20 | 0x40, 3)):
                args = [px, w, h, fmt, w * B, flags] + ([alpha_mode] if alpha_mode >= 0 else [])
                src = os.path.join(str(tmp_path), "px.bin"); out = os.path.join(str(tmp_path), "out.tga")
                px.tofile(src)
                cmd = [EXE, "codec_save", "tga", src, str(w), str(h), str(fmt), str(w * B), str(flags), out] + ([str(alpha_mode)] if alpha_mode >= 0 else [])
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=60)
                hr = int(r.stdout.split()[1], 16)
                rhr, ref = oracle.ref_save_tga(px, w, h, fmt, w * B, flags, alpha_mode)
                assert hr == rhr, (fmt, flags, alpha_mode, hex(hr), hex(rhr))
                if ref is None:
                    continue
                ours = np.fromfile(out, np.uint8)
                assert ours.size == ref.size
                if alpha_mode >= 0:
                    at = ref.size - 26 - 495
                    ours[at + TGA_STAMP.start:at + TGA_STAMP.stop] = 0; ref = ref.copy(); ref[at + TGA_STAMP.start:at + TGA_STAMP.stop] = 0
                assert np.array_equal(ours, ref), (fmt, flags, alpha_mode)
                n += 1
    assert n > 200
    # padded source rows
    px = rng.integers(0, 256, (3, 40), dtype=np.uint8)
    src = os.path.join(str(tmp_path), "px.bin"); out = os.path.join(str(tmp_path), "out.tga")
    px.tofile(src)
    subprocess.run([EXE, "codec_save", "tga", src, "5", "3", "28", "40", "0", out], check=True, capture_output=True)
    assert np.array_equal(np.fromfile(out, np.uint8), oracle.ref_save_tga(px, 5, 3, 28, 40)[1])


def test_tga_seeded_mutations(tmp_path):
    rng = np.random.default_rng(23)
    bases = [f for f in tga_files(np.random.default_rng(5)) if len(f) > 40][:160]
    for flags in (0, 0x1 | 0x80):
        files = []
        for i in range(1500):
            b = bytearray(bases[int(rng.integers(len(bases)))])
            for _ in range(int(rng.integers(1, 4))):
                r = rng.random()
                at = int(rng.integers(0, 18)) if r < 0.5 else (len(b) - 1 - int(rng.integers(0, min(len(b), 540))) if r < 0.7 else int(rng.integers(0, len(b))))
                b[at] = int(rng.choice([0, 1, 2, 3, 8, 9, 10, 11, 15, 16, 24, 32, 127, 128, 255, int(rng.integers(0, 256))]))
            if rng.random() < 0.15:
                del b[int(rng.integers(0, len(b))):]
            files.append(bytes(b))
        tga_compare(str(tmp_path), files, flags, "mutations")
