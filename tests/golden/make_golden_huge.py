#!/usr/bin/env python3
"""Golden digests of the reference on images of the LARGEST sizes the path has to take (tests/golden/huge.json).

    python tests/golden/make_golden_huge.py            # a few minutes on 8 cores, ~3 GiB of host memory

DirectXTexImage.cpp:127-131 lets a ScratchImage hold 16384 x 16384 textures; the BC6H / BC7 search pipeline works in passes of at most
2^22 blocks (csrc/search_common.h: build_passes). Until round 4 both had only run shrunk (a 52 x 36 image with a 17-block pass). The
cases here are full-size:
  * huge_bc1_16384: 16384 x 16384 RGBA8 (1 GiB, 16 777 216 blocks, rows of 64 KiB) -> BC1;
  * huge_bc7_passes: 16380 x 4102 RGBA8 -> BC7: 4095 x 1026 = 4 201 470 blocks > 2^22, so the product's own pass size cuts it in two
    passes, the cut falls in the middle of block row 1024 (2^22 = 1024 x 4095 + 1024), and the last block row is partial (2 texel rows).
The images are directxtex_amd.synth.huge_rgba8: flat blocks (which the reference leaves through its early-outs) with SURVEY-recipe
strips where the seams are, so the reference finishes in minutes. Stored: SHA-256 of the payload and of bands of 16 block rows."""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RGBA8, BC1, BC7 = 28, 71, 98
TEX_COMPRESS_PARALLEL = 0x10000000
BAND_ROWS = 16

# (id, width, height, seed, noisy block rows, BC format)
CASES = [
    ("huge_bc1_16384", 16384, 16384, 41, [0, 1, 2047, 2048, 2049, 4094, 4095], BC1),
    ("huge_bc7_passes", 16380, 4102, 42, [0, 1023, 1024, 1025], BC7),
]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def band_digests(payload, width, height, block_bytes):
    nbw, nbh = (width + 3) // 4, (height + 3) // 4
    rows = np.ascontiguousarray(payload, np.uint8).reshape(nbh, nbw * block_bytes)
    return [sha(rows[r:r + BAND_ROWS]) for r in range(0, nbh, BAND_ROWS)]


def make_input(case):
    from directxtex_amd import synth
    cid, w, h, seed, noisy, fmt = case
    return synth.huge_rgba8(w, h, seed, noisy, "opaque")


def main():
    import oracle
    assert oracle.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    out_path = os.path.join(HERE, "huge.json")
    gold = json.load(open(out_path)) if os.path.exists(out_path) else {"band_rows": BAND_ROWS, "cases": {}}
    only = set(sys.argv[1:])
    for case in CASES:
        cid, w, h, seed, noisy, fmt = case
        if (only and cid not in only) or (not only and cid in gold["cases"]):
            continue
        img = make_input(case)
        t0 = time.perf_counter()
        pay = oracle.ref_compress_image(img, w, h, RGBA8, fmt, TEX_COMPRESS_PARALLEL, 0.5)
        dt = time.perf_counter() - t0
        bb = 8 if fmt == BC1 else 16
        gold["cases"][cid] = {"width": w, "height": h, "blocks": ((w + 3) // 4) * ((h + 3) // 4), "input_sha256": sha(img), "bytes": int(pay.nbytes),
                              "sha256": sha(pay), "bands": band_digests(pay, w, h, bb), "ref_seconds": round(dt, 1), "ref_threads": oracle.ref_num_threads()}
        print(cid, f"{dt:.1f} s", flush=True)
        json.dump(gold, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
