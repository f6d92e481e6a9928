#!/usr/bin/env python3
"""DEVELOPMENT TOOL: one BC7 encode of a size x size synthetic image, for rocprofv3 runs (kernel trace / PMC passes).
usage: python tools/prof_bc7.py [size] [format] [flags]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import directxtex_amd as dx
from directxtex_amd import synth

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
fmt = int(sys.argv[2]) if len(sys.argv) > 2 else dx.DXGI_FORMAT_BC7_UNORM
flags = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
ctx = dx.Context(0)
img = synth.rgba8(size, size, seed=32, alpha="opaque")
out = ctx.compress(img, size, size, dx.DXGI_FORMAT_R8G8B8A8_UNORM, fmt, flags, 0.5)
print("kernel ms", ctx.last_kernel_ms(), "bytes", out.size)
ctx.close()
