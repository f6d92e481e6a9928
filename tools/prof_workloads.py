#!/usr/bin/env python3
"""DEVELOPMENT TOOL (run under rocprofv3 on the GPU box): executes BASELINE.json's workloads a fixed number of times so that a
kernel trace / PMC pass attributes counters to their kernels.
usage: python tools/prof_workloads.py bc7|others [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import ctypes
import numpy as np, torch
import directxtex_amd as dx
from directxtex_amd import synth

what = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = dx.Context(0); dev = torch.device("cuda", 0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
RGBA8, RGBA16F = 28, 10
W = H = 4096
img = synth.survey_rgba8(W, H, 2, "opaque")
src = torch.from_numpy(img).to(dev)


def out_for(fmt, w=W, h=H):
    return torch.empty(dx.compute_pitch(fmt, w, h)[1], dtype=torch.uint8, device=dev)


if what == "bc7":
    dst = out_for(98)
    for _ in range(reps):
        ctx.compress_device(src.data_ptr(), W, H, RGBA8, dst.data_ptr(), 98, 0, 0.5)
    torch.cuda.synchronize()
else:
    for fmt in (71, 77, 83):                                      # BC1, BC3, BC5
        dst = out_for(fmt)
        for _ in range(reps):
            ctx.compress_device(src.data_ptr(), W, H, RGBA8, dst.data_ptr(), fmt, 0, 0.5)
    hdr = torch.from_numpy(synth.survey_rgba16f(W, H, 3)).to(dev)  # cfg3
    dst = out_for(95)
    for _ in range(reps):
        ctx.compress_device(hdr.data_ptr(), W, H, RGBA16F, dst.data_ptr(), 95, 0, 0.5)
    bc7 = out_for(98)                                              # decode: BC7 (arbitrary blocks: every mode), BC1, BC3, BC6H
    bc7.copy_(torch.randint(0, 256, (bc7.numel(),), dtype=torch.uint8, device=dev))
    bc1 = out_for(71); bc3_ = out_for(77)
    ctx.compress_device(src.data_ptr(), W, H, RGBA8, bc1.data_ptr(), 71, 0, 0.5)
    ctx.compress_device(src.data_ptr(), W, H, RGBA8, bc3_.data_ptr(), 77, 0, 0.5)
    back = torch.empty(W * H * 4, dtype=torch.uint8, device=dev)
    for _ in range(reps):
        ctx.decompress_device(bc7.data_ptr(), W, H, 98, back.data_ptr(), RGBA8)
        ctx.decompress_device(bc1.data_ptr(), W, H, 71, back.data_ptr(), RGBA8)
        ctx.decompress_device(bc3_.data_ptr(), W, H, 77, back.data_ptr(), RGBA8)
        ctx.decompress_device(dst.data_ptr(), W, H, 95, hdr.data_ptr(), RGBA16F)
    cv = torch.empty(W * H * 8, dtype=torch.uint8, device=dev)     # convert RGBA8 -> RGBA16F
    s_im = dx.capi.device_image(src.data_ptr(), W, H, RGBA8); d_im = dx.capi.device_image(cv.data_ptr(), W, H, RGBA16F)
    for _ in range(reps):
        ctx._check(dx.capi._lib.dxtex_convert_device(ctx._h, ctypes.byref(s_im), ctypes.byref(d_im), 0, 0.5), "convert_device")
    big = torch.from_numpy(synth.survey_rgba8(8192, 8192, 4, "random")).to(dev)   # cfg4
    sizes = []; w = h = 8192
    while True:
        sizes.append((w, h))
        if w == 1 and h == 1: break
        w, h = max(1, w >> 1), max(1, h >> 1)
    bufs = [big.reshape(-1)] + [torch.empty(a * b * 4, dtype=torch.uint8, device=dev) for a, b in sizes[1:]]
    levels = [dx.capi.device_image(t.data_ptr(), a, b, RGBA8) for t, (a, b) in zip(bufs, sizes)]
    for _ in range(reps):
        ctx.generate_mips_device(levels, dx.TEX_FILTER_CUBIC)
    for _ in range(reps):
        ctx.generate_mips_device(levels, dx.TEX_FILTER_BOX)
    bc3 = [out_for(77, a, b) for a, b in sizes]
    dsts = [dx.capi.device_image(t.data_ptr(), a, b, 77) for t, (a, b) in zip(bc3, sizes)]
    for _ in range(reps):
        ctx.compress_many_device(levels, dsts, 0, 0.5)
    torch.cuda.synchronize()
print("done", what)
