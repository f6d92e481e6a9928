#!/usr/bin/env python3
"""DEVELOPMENT TOOL: one lone BC7 image of side S (default 512), a few repetitions - for a rocprofv3 --kernel-trace timeline of the small-pass plan.
usage: python tools/small_probe.py [--dev] [S] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import directxtex_amd as dx
if "--dev" in sys.argv:
    sys.argv.remove("--dev"); dx.capi.load(dev=True)
from directxtex_amd import synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = dx.Context(0)
img = synth.survey_rgba8(S, S, 2, "opaque")
n = dx.compute_pitch(98, S, S)[1]
src = ctx.device_alloc(img.nbytes); dst = ctx.device_alloc(n)
ctx.upload(src, img, sync=True)
for _ in range(reps):
    ctx.compress_device(src, S, S, 28, dst, 98, 0, 0.5)
    ctx.synchronize()
print("done")
