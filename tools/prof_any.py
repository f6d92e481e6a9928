#!/usr/bin/env python3
"""DEVELOPMENT TOOL: times one workload with the per-kernel profiler of the C ABI.
usage: python tools/prof_any.py bc6h|bc3|bc1|mips_box|mips_cubic|convert|decode [size]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import directxtex_amd as dx
from directxtex_amd import synth
what = sys.argv[1]; size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ctx = dx.Context(0); dev = torch.device("cuda", 0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
rgba = synth.rgba8(min(size, 1024), min(size, 1024), seed=5, alpha="smooth")
reps = size // rgba.shape[0]
rgba = np.tile(rgba, (reps, reps, 1))
def run(fn, n=3):
    fn(); torch.cuda.synchronize()
    ctx.profile_begin(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    k = ctx.profile_end()
    return dt, {a: round(ms / max(1, c) , 4) for a, (ms, c) in k.items()}
if what in ("bc6h",):
    img = torch.from_numpy((rgba.astype(np.float32) / 255.0 * 4.0).astype(np.float16)).to(dev)
    rp, sp = dx.compute_pitch(95, size, size); out = torch.empty(sp, dtype=torch.uint8, device=dev)
    dt, k = run(lambda: ctx.compress_device(img.data_ptr(), size, size, 10, out.data_ptr(), 95, 0, 0.5), 2)
    print("BC6H_UF16 %dx%d: %.2f ms, %.1f Mtexels/s" % (size, size, dt * 1e3, size * size / dt / 1e6)); print(k)
elif what in ("bc1", "bc3", "bc5"):
    fmt = {"bc1": 71, "bc3": 77, "bc5": 83}[what]
    img = torch.from_numpy(rgba).to(dev)
    rp, sp = dx.compute_pitch(fmt, size, size); out = torch.empty(sp, dtype=torch.uint8, device=dev)
    dt, k = run(lambda: ctx.compress_device(img.data_ptr(), size, size, 28, out.data_ptr(), fmt, 0, 0.5), 10)
    bpt = 4.5 if fmt == 71 else 5.0
    print("%s %dx%d: %.3f ms, %.0f Mtexels/s, %.0f GB/s algorithmic" % (what, size, size, dt * 1e3, size * size / dt / 1e6, size * size * bpt / dt / 1e9))
