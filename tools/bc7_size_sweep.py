#!/usr/bin/env python3
"""DEVELOPMENT TOOL: BC7 encode rate against image size (tail / launch effects of the per-mode pipeline).
usage: python tools/bc7_size_sweep.py [sizes...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import directxtex_amd as dx
from directxtex_amd import synth
sizes = [int(s) for s in sys.argv[1:]] or [512, 1024, 2048, 4096, 8192]
ctx = dx.Context(0); dev = torch.device("cuda", 0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
base = synth.rgba8(1024, 1024, seed=5, alpha="opaque")
for size in sizes:
    reps = max(1, size // 1024)
    img = torch.from_numpy(np.tile(base, (reps, reps, 1))[:size, :size].copy()).to(dev)
    rp, sp = dx.compute_pitch(98, size, size); out = torch.empty(sp, dtype=torch.uint8, device=dev)
    fn = lambda: ctx.compress_device(img.data_ptr(), size, size, 28, out.data_ptr(), 98, 0, 0.5)
    fn(); torch.cuda.synchronize()
    n = 2 if size >= 4096 else 4
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("BC7 %5d^2: %9.2f ms  %6.2f Mtexels/s" % (size, dt * 1e3, size * size / dt / 1e6), flush=True)
