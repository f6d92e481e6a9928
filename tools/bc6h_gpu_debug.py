#!/usr/bin/env python3
"""DEVELOPMENT TOOL (GPU box): dumps tiles whose BC6H encode differs from the reference, plus the GPU's per-mode result."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import directxtex_amd as dx
import oracle
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_bc6h_parity import _hdr_image
fmt = int(sys.argv[1]) if len(sys.argv) > 1 else 95
w = h = 64
img = _hdr_image(w, h, seed=w + h + fmt, signed=(fmt == 96)).astype(np.float32)
tiles = oracle.gather_tiles(img)
ctx = dx.Context(0)
got = ctx.encode_blocks(fmt, tiles, 0); ref = oracle.ref_encode_blocks(fmt, tiles, 0)
bad = np.nonzero((got != ref).any(axis=1))[0]
print("bad", bad[:20])
sel = tiles[bad[:4]]
sel.astype(np.float32).tofile(os.path.join(ROOT, "gpurun_out", "bc6h_bad_tiles.bin"))
for t in range(len(sel)):
    print("tile", bad[t], "ref", ref[bad[t]].tobytes()[::-1].hex(), "gpu", got[bad[t]].tobytes()[::-1].hex())
ctx.close()
for mode in range(14):
    for ns in (0, 1):
        env = dict(os.environ, DXTEX_BC6H_ONLY_MODE=str(mode))
        if ns: env["DXTEX_BC6H_NO_SEARCH"] = "1"
        code = ("import sys; sys.path.insert(0, %r); import numpy as np, directxtex_amd as dx; dx.capi.load(dev=True); c = dx.Context(0); "
                "t = np.fromfile(%r, np.float32).reshape(-1,16,4); o = c.encode_blocks(%d, t, 0); "
                "print(' '.join(b.tobytes()[::-1].hex() for b in o))") % (ROOT, os.path.join(ROOT, "gpurun_out", "bc6h_bad_tiles.bin"), fmt)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print("mode", mode, "nosearch", ns, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
