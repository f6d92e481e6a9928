#!/usr/bin/env python3
"""Golden digests of the reference for BASELINE.json's cfg5 (tests/golden/cfg5.json).

cfg5 is 1024 x 2048^2 RGBA8 -> BC7; SURVEY.md section 8d cycles 16 distinct host images (directxtex_amd.synth.survey_rgba8,
LCG seeds 1000 .. 1015, opaque), image i of the array = distinct image i mod 16. Every payload here is the reference's own
DirectX::Compress (TEX_COMPRESS_PARALLEL -> CompressBC_Parallel, DirectXTexCompress.cpp:210-372, D3DXEncodeBC7) compiled in
place into oracle/_ref. Run in the container that holds /root/reference:
    python tests/golden/make_golden_cfg5.py            # ~40 min on 8 cores
bench.py's cfg5 leg and tests/test_many_gpu.py compare dxtex_compress_many's payloads with these digests (no oracle in the loop).
"""
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

SIDE = 2048
SEED0 = 1000
DISTINCT = 16
TEX_COMPRESS_PARALLEL = 0x10000000


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    import oracle
    from directxtex_amd import synth
    assert oracle.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    out_path = os.path.join(HERE, "cfg5.json")
    gold = {"side": SIDE, "seed0": SEED0, "distinct": DISTINCT, "alpha": "opaque", "images": {}}
    if os.path.exists(out_path):
        gold = json.load(open(out_path))
    for k in range(DISTINCT):
        if str(k) in gold["images"]:
            continue
        img = synth.survey_rgba8(SIDE, SIDE, SEED0 + k, "opaque")
        t0 = time.perf_counter()
        pay = oracle.ref_compress_image(img, SIDE, SIDE, 28, 98, TEX_COMPRESS_PARALLEL, 0.5)
        dt = time.perf_counter() - t0
        gold["images"][str(k)] = {"input_sha256": sha(img), "sha256": sha(pay), "ref_seconds": round(dt, 2), "ref_threads": oracle.ref_num_threads()}
        print(k, f"{dt:.1f} s", flush=True)
        json.dump(gold, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
