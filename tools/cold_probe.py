#!/usr/bin/env python3
"""DEVELOPMENT TOOL (asked for by the round-5 review): what the FIRST BC7 call of a fresh process and a fresh context costs, with and without
dxtex_ctx_prepare - Texconv compresses one file per call (Texconv/texconv.cpp:3692-3712), so a cold context is what it would meet.
Run once per variant in a NEW process (code objects are loaded lazily by the HIP runtime on the first launch of each kernel):
    python tools/cold_probe.py            # no prepare
    python tools/cold_probe.py prepare    # dxtex_ctx_prepare(64, 64, RGBA8 -> BC7) before the first call
Prints the wall time of context creation, of prepare, of the first three host-to-host compress calls of a 64 x 64 image and of the first
512 x 512 call, and whether the payloads equal the oracle's when it is built (checker only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
t0 = time.perf_counter()
import numpy as np
import directxtex_amd as dx
from directxtex_amd import synth
t_import = time.perf_counter() - t0

RGBA8, BC7 = 28, 98
img64 = synth.survey_rgba8(64, 64, 2, "opaque")
img512 = synth.survey_rgba8(512, 512, 2, "opaque")

t0 = time.perf_counter(); ctx = dx.Context(0); t_ctx = time.perf_counter() - t0
t_prep = None
if "prepare" in sys.argv[1:]:
    t0 = time.perf_counter(); ctx.prepare(64, 64, RGBA8, BC7); t_prep = time.perf_counter() - t0
times = []
out64 = None
for _ in range(3):
    t0 = time.perf_counter(); out64 = ctx.compress(img64, 64, 64, RGBA8, BC7); times.append(time.perf_counter() - t0)
t0 = time.perf_counter(); out512 = ctx.compress(img512, 512, 512, RGBA8, BC7); t512a = time.perf_counter() - t0
t0 = time.perf_counter(); out512 = ctx.compress(img512, 512, 512, RGBA8, BC7); t512b = time.perf_counter() - t0
print("cold probe (%s): import %.1f ms, context %.1f ms%s; 64x64 BC7 host-to-host calls 1 / 2 / 3: %.2f / %.2f / %.2f ms; 512x512 first / second: %.2f / %.2f ms" % (
    "with dxtex_ctx_prepare" if t_prep is not None else "no prepare", t_import * 1e3, t_ctx * 1e3,
    ", prepare %.2f ms" % (t_prep * 1e3) if t_prep is not None else "", times[0] * 1e3, times[1] * 1e3, times[2] * 1e3, t512a * 1e3, t512b * 1e3))
try:
    import oracle
    if oracle.have_ref():
        ref = oracle.ref_compress_image(img64, 64, 64, RGBA8, BC7, 0)
        print("  64x64 payload", "identical to the reference" if bytes(ref) == bytes(out64) else "DIFFERS from the reference")
except Exception as e:      # the checker is optional here
    print("  (oracle not available: %s)" % e)
ctx.close()
