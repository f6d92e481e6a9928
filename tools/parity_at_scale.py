#!/usr/bin/env python3
"""DEVELOPMENT TOOL: GPU payloads against the reference's CPU encoder (oracle/_ref, OpenMP) on images large enough to hold
hundreds of thousands of blocks of every kind. Writes a markdown summary. usage: python tools/parity_at_scale.py [side] [out.md]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import directxtex_amd as dx
from directxtex_amd import synth
import oracle

side = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
out_md = sys.argv[2] if len(sys.argv) > 2 else None
ctx = dx.Context(0)
q = side // 4
yy, xx = np.mgrid[0:side, 0:side]
mix = synth.rgba8(side, side, seed=77, alpha="opaque")
mix[:q * 2, q * 2:] = synth.rgba8(q * 2, side - q * 2, seed=78, alpha="smooth")                 # noise with smooth alpha
mix[q * 2:, :q * 2, :3] = np.stack([(xx[q * 2:, :q * 2] // 3) % 256, (yy[q * 2:, :q * 2] // 2) % 256, ((xx + yy)[q * 2:, :q * 2] // 5) % 256], -1)   # gradients
mix[q * 3:, q * 3:] = np.random.default_rng(9).integers(0, 256, (side - q * 3, side - q * 3, 4), dtype=np.uint8)                # pure RGBA noise
mix[q * 2:q * 3, q * 2:q * 3] = (mix[q * 2:q * 3, q * 2:q * 3] // 64) * 64                                                     # flat patches
hdr = (mix[..., :4].astype(np.float32) / 255.0 * 12.0).astype(np.float16)
hdr[:q] *= np.float16(0.01)
rows = []
for name, img, sfmt, dfmt, flags in (("BC7 DEFAULT", mix, 28, 98, 0), ("BC7 USE_3SUBSETS", mix[:side // 2, :side // 2], 28, 98, 0x80000),
                                     ("BC6H_UF16", hdr, 10, 95, 0), ("BC6H_SF16", (hdr - np.float16(3.0)), 10, 96, 0), ("BC3", mix, 28, 77, 0), ("BC1 dither", mix, 28, 71, 0x10000)):
    h, w = img.shape[:2]
    img = np.ascontiguousarray(img)
    t0 = time.perf_counter(); got = ctx.compress(img, w, h, sfmt, dfmt, flags, 0.5); tg = time.perf_counter() - t0
    t0 = time.perf_counter(); ref = oracle.ref_compress_image(img, w, h, sfmt, dfmt, flags, 0.5); tr = time.perf_counter() - t0
    bb = dx.BC_BLOCK_BYTES[dfmt]
    same = (got.reshape(-1, bb) == ref.reshape(-1, bb)).all(axis=1)
    rows.append((name, w, h, same.size, int(same.sum()), tg, tr))
    print("%-18s %dx%d: %d of %d blocks identical (GPU incl. transfers %.2f s, serial reference %.1f s)" % (name, w, h, same.sum(), same.size, tg, tr), flush=True)
if out_md:
    with open(out_md, "w") as f:
        f.write("# Parity at scale: GPU payload vs the reference's CPU encoder (oracle/_ref), mixed-content image\n\n")
        f.write("`python tools/parity_at_scale.py %d` on the MI355X box (reference: `DirectX::Compress` without\n`TEX_COMPRESS_PARALLEL`, i.e. its serial `CompressBC` loop). The image mixes opaque noise, noise with\nsmooth alpha, gradients, pure RGBA noise and flat patches, so every mode, the pruning and the phase scheduling are exercised.\n\n" % side)
        f.write("| encode | image | blocks | identical to the reference | GPU (host buffers) | reference |\n|---|---|---|---|---|---|\n")
        for name, w, h, n, k, tg, tr in rows:
            f.write("| %s | %d x %d | %d | %d (%.4f %%) | %.2f s | %.1f s |\n" % (name, w, h, n, k, 100.0 * k / n, tg, tr))
assert all(r[3] == r[4] for r in rows), "mismatch"
