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
elif what == "decode":
    img = torch.from_numpy(rgba).to(dev)
    for name, fmt, dstfmt, flags in (("bc1", 71, 28, 0), ("bc3", 77, 28, 0), ("bc5", 83, 28, 0), ("bc7", 98, 28, 0x100000), ("bc6h", 95, 10, 0)):
        srcfmt, src = 28, img
        if fmt == 95:
            srcfmt, src = 10, torch.from_numpy((rgba.astype(np.float32) / 255.0 * 4.0).astype(np.float16)).to(dev)
            if size > 2048:
                pass
        rp, sp = dx.compute_pitch(fmt, size, size); bc = torch.empty(sp, dtype=torch.uint8, device=dev)
        ctx.compress_device(src.data_ptr(), size, size, srcfmt, bc.data_ptr(), fmt, flags, 0.5)
        bpp = 8 if dstfmt == 10 else 4
        back = torch.empty(size * size * bpp, dtype=torch.uint8, device=dev)
        dt, k = run(lambda: ctx.decompress_device(bc.data_ptr(), size, size, fmt, back.data_ptr(), dstfmt), 20)
        by = sp + size * size * bpp
        print("decode %s %dx%d: %.3f ms, %.0f Mtexels/s, %.0f GB/s algorithmic  %s" % (name, size, size, dt * 1e3, size * size / dt / 1e6, by / dt / 1e9, k))
elif what in ("mips_box", "mips_cubic", "mips_linear", "mips_triangle"):
    flt = {"mips_box": dx.TEX_FILTER_BOX, "mips_cubic": dx.TEX_FILTER_CUBIC, "mips_linear": dx.TEX_FILTER_LINEAR, "mips_triangle": dx.TEX_FILTER_TRIANGLE}[what]
    img = torch.from_numpy(rgba).to(dev)
    sizes = []; w = h = size
    while True:
        sizes.append((w, h))
        if w == 1 and h == 1: break
        w, h = max(1, w >> 1), max(1, h >> 1)
    bufs = [img.reshape(-1)] + [torch.empty(a * b * 4, dtype=torch.uint8, device=dev) for a, b in sizes[1:]]
    levels = [dx.capi.device_image(t.data_ptr(), a, b, 28) for t, (a, b) in zip(bufs, sizes)]
    by = sum(a * b * 4 for a, b in sizes[:-1]) + sum(a * b * 4 for a, b in sizes[1:])
    dt, k = run(lambda: ctx.generate_mips_device(levels, flt), 10)
    print("%s %dx%d: %.3f ms, %.0f GB/s algorithmic %s" % (what, size, size, dt * 1e3, by / dt / 1e9, k))
