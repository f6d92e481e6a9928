#!/usr/bin/env python3
"""DEVELOPMENT TOOL: per-kernel times (C ABI profiler) of a small BC7 encode. usage: python tools/prof_bc7_small.py [size]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import directxtex_amd as dx
from directxtex_amd import synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = dx.Context(0); dev = torch.device("cuda", 0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
img = torch.from_numpy(synth.rgba8(size, size, seed=5, alpha="opaque")).to(dev)
rp, sp = dx.compute_pitch(98, size, size); out = torch.empty(sp, dtype=torch.uint8, device=dev)
fn = lambda: ctx.compress_device(img.data_ptr(), size, size, 28, out.data_ptr(), 98, 0, 0.5)
fn(); torch.cuda.synchronize()
t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); print("wall ms", (time.perf_counter() - t0) * 1e3)
ctx.profile_begin(); fn(); torch.cuda.synchronize(); k = ctx.profile_end()
tot = 0
for name, (ms, c) in sorted(k.items(), key=lambda kv: -kv[1][0]):
    print("%-34s %8.3f ms x%d" % (name, ms, c)); tot += ms
print("sum", tot)
