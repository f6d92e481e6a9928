#!/usr/bin/env python3
"""Generates tests/golden/codecs.json: what THE REFERENCE's readers (oracle/_ref: DirectXTexDDS.cpp, DirectXTexHDR.cpp,
DirectXTexTGA.cpp compiled in place) return for a fixed, seeded set of DDS / HDR / TGA files - HRESULT, metadata and a SHA-256 of
the pixels. The files themselves are regenerated from the seeds by tests/test_golden_codecs_cpu.py (same functions, imported
from here), so only digests are stored. Run where /root/reference exists:   python tests/golden/make_golden_codecs.py
The test then checks the host layer's readers against these digests with no oracle in the loop."""
import hashlib, json, os, struct, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def dds_files():
    """[(name, flags, bytes)]: every legacy pixel format under a few reader flags, plus 'DX10' files of several dimensions."""
    import test_dds_legacy_cpu as L
    rng = np.random.default_rng(2024)
    payload = rng.integers(0, 256, 1 << 14, dtype=np.uint8).tobytes()
    out = []
    for i, pf in enumerate(L.LEGACY):
        for fl in (0, 0x10, 0x20 | 0x8, 0x1 | 0x4):
            out.append((f"dds/legacy{i}/{fl:#x}", fl, L.header(9, 6, pf, mips=2) + payload))
    dx = L.four("DX10")
    for ext in ((28, 3, 0, 2, 0), (28, 3, 4, 1, 3), (71, 3, 0, 1, 1), (98, 4, 0, 1, 0), (10, 2, 0, 3, 0), (85, 3, 0, 1, 0), (115, 3, 0, 1, 0), (24, 3, 0, 1, 0)):
        for fl in (0, 0x8 | 0x10, 0x100):
            hf = 0x1007 | (0x800000 if ext[1] == 4 else 0)
            out.append((f"dds/dx10-{'-'.join(map(str, ext))}/{fl:#x}", fl, L.header(8, 1 if ext[1] == 2 else 8, dx, flags=hf, depth=4 if ext[1] == 4 else 0, mips=3, dx10=ext) + payload))
    return out


def hdr_files():
    rng = np.random.default_rng(2025)
    out = []
    for k, (w, h) in enumerate(((16, 4), (9, 3), (130, 2))):
        body = bytes(int(v) for v in rng.integers(3, 256, w * h * 4))
        out.append((f"hdr/raw{k}", 0, f"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y {h} +X {w}\n".encode() + body))
        out.append((f"hdr/exposure{k}", 0, f"#?RGBE\nEXPOSURE=2.5\nFORMAT=32-bit_rle_xyze\nEXPOSURE= 0.25\n\n-Y {h} +X {w}\n".encode() + body))
        # new-scheme run-length rows: per channel, runs of 5 and literals of 3
        rows = bytearray()
        for _ in range(h):
            rows += bytes([2, 2, w >> 8, w & 255])
            for c in range(4):
                x = 0
                while x < w:
                    n = min(5, w - x)
                    if (x // 5) % 2 == 0:
                        rows += bytes([128 + n, int(rng.integers(0, 256))])
                    else:
                        rows += bytes([n]) + bytes(int(v) for v in rng.integers(0, 256, n))
                    x += n
        out.append((f"hdr/rle{k}", 0, f"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y {h} +X {w}\n".encode() + bytes(rows)))
        out.append((f"hdr/truncated{k}", 0, f"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y {h} +X {w}\n".encode() + bytes(rows)[:len(rows) // 2]))
    out.append(("hdr/bad-orientation", 0, b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n+Y 3 +X 5\n" + bytes(60)))
    out.append(("hdr/no-format", 0, b"#?RADIANCE\n\n-Y 3 +X 5\n" + bytes(60)))
    return out


def tga_files():
    import test_hdr_tga_cpu as T
    files = T.tga_files(np.random.default_rng(2026))
    out = []
    for i, f in enumerate(files):
        for fl in (0, 0x1 | 0x80):
            out.append((f"tga/{i}/{fl:#x}", fl, bytes(f)))
    return out


def digest(hr, meta, px):
    return {"hr": f"{hr:08x}", "meta": meta, "sha256": hashlib.sha256(px.tobytes()).hexdigest() if px is not None else None}


def main():
    import oracle
    cases = {}
    for name, fl, data in dds_files():
        hr, meta, px = oracle.ref_load_dds_ex(np.frombuffer(data, np.uint8), fl, capacity=1 << 22)
        cases[name] = digest(hr, meta, px)
    for name, fl, data in hdr_files():
        hr, meta, px = oracle.ref_load_hdr(data)
        cases[name] = digest(hr, meta, px)
    for name, fl, data in tga_files():
        hr, meta, px = oracle.ref_load_tga(data, fl)
        cases[name] = digest(hr, meta, px)
    path = os.path.join(ROOT, "tests", "golden", "codecs.json")
    with open(path, "w") as f:
        json.dump({"_generator": "tests/golden/make_golden_codecs.py (the reference's readers through oracle/_ref)", "cases": cases}, f, indent=0, sort_keys=True)
    print(f"{len(cases)} cases -> {path} ({os.path.getsize(path)} bytes), loaded {sum(1 for c in cases.values() if c['sha256'])}")


if __name__ == "__main__":
    main()
