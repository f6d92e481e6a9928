#!/usr/bin/env python3
"""DEVELOPMENT TOOL: torch-free A/B probe (device buffers through libamdhip64 directly, so a fresh GPU box does not pay for importing
torch): wall time per image, the context's per-kernel times and the golden check of the full-size BC7 (cfg2) / BC6H (cfg3) encodes.
usage: [DXTEX_...=...] python tools/quick_probe.py [--dev] [bc7] [bc6h] [bc1] [small] [cfg4]
(round 4: payload digests for every run, so A/B variants can be compared byte for byte; cfg4 = the 8192^2 mip chains + BC3 of the chain against the golden digests)"""
import ctypes, hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import directxtex_amd as dx
if "--dev" in sys.argv:
    sys.argv.remove("--dev"); dx.capi.load(dev=True)      # the -DDXTEX_DEV build: the only one that reads DXTEX_* knobs
from directxtex_amd import synth

hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so", mode=os.RTLD_LAZY)      # lazy: the image's HSA runtime lacks a symbol RTLD_NOW insists on
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipFree.argtypes = [ctypes.c_void_p]


def dmalloc(n):
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), n) == 0
    return p.value


def h2d(arr):
    arr = np.ascontiguousarray(arr)
    p = dmalloc(arr.nbytes)
    assert hip.hipMemcpy(p, arr.ctypes.data, arr.nbytes, 1) == 0
    return p


def d2h(p, n):
    out = np.empty(n, np.uint8)
    assert hip.hipMemcpy(out.ctypes.data, p, n, 2) == 0
    return out


def cached(name, make):
    """synthetic images are a few seconds of numpy each: several probe runs in one gpurun call share them through /tmp"""
    path = "/tmp/dxtex_probe_%s.npy" % name
    if os.path.exists(path):
        return np.load(path)
    img = make(); np.save(path, img)
    return img


what = set(sys.argv[1:]) or {"bc7"}
TOP = int(os.environ.get("PROBE_TOP", "16"))
ctx = dx.Context(0)
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json")))["cases"]


def run(name, img, sfmt, dfmt, W, H, gold=None, reps=3):
    reps = int(os.environ.get("PROBE_REPS", reps))
    src = h2d(img)
    nbytes = dx.compute_pitch(dfmt, W, H)[1]
    dst = dmalloc(nbytes)
    f = lambda: ctx.compress_device(src, W, H, sfmt, dst, dfmt, 0, 0.5)
    f(); ctx.synchronize()
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(reps): f()
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    print("%s: %.3f ms per image (wall), %.1f Mtexels/s" % (name, best * 1e3, W * H / best / 1e6))
    ctx.profile_begin(); f(); k = ctx.profile_end()
    print("  serial kernel sum %.2f ms" % sum(ms for ms, n in k.values()))
    for kn, (ms, n) in sorted(k.items(), key=lambda kv: -kv[1][0])[:TOP]:
        print("  %-40s %8.3f ms x%d" % (kn, ms, n))
    sha = hashlib.sha256(d2h(dst, nbytes).tobytes()).hexdigest()
    if gold:
        print("  payload", "IDENTICAL to the reference golden" if sha == GOLD[gold]["sha256"] else "DIFFERS from the reference golden")
    else:
        print("  payload sha256", sha[:16])
    hip.hipFree(src); hip.hipFree(dst)


if "bc7" in what:
    run("bc7 4096^2 cfg2", cached("cfg2", lambda: synth.survey_rgba8(4096, 4096, 2, "opaque")), 28, 98, 4096, 4096, "cfg2_bc7_4096")
if "bc6h" in what:
    run("bc6h 4096^2 cfg3", cached("cfg3", lambda: synth.survey_rgba16f(4096, 4096, 3)), 10, 95, 4096, 4096, "cfg3_bc6h_uf16_4096", reps=2)
if "bc1" in what:
    img = cached("cfg2", lambda: synth.survey_rgba8(4096, 4096, 2, "opaque"))
    for fmt, nm in ((71, "bc1"), (74, "bc2"), (77, "bc3"), (80, "bc4"), (83, "bc5")):
        run(nm + " 4096^2", img, 28, fmt, 4096, 4096, None, reps=20)
if "small" in what:
    for S in [int(x) for x in os.environ.get("PROBE_SIZES", "256,512,1024").split(",")]:
        run("bc7 %d^2" % S, synth.survey_rgba8(S, S, 2, "opaque"), 28, 98, S, S, None, reps=5)
if "cfg4" in what:
    W = H = 8192
    img = synth.survey_rgba8(W, H, 4, "random")
    sizes = []
    w, h = W, H
    while True:
        sizes.append((w, h))
        if w == 1 and h == 1: break
        w, h = max(1, w >> 1), max(1, h >> 1)
    bufs = [h2d(img)] + [dmalloc(a * b * 4) for a, b in sizes[1:]]
    levels = [dx.capi.device_image(p, a, b, 28) for p, (a, b) in zip(bufs, sizes)]
    for name, flt in (("cubic", 0x300000), ("box", 0x400000)):
        f = lambda: ctx.generate_mips_device(levels, flt)
        f(); ctx.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(10): f()
            ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) / 10)
        got = [hashlib.sha256(d2h(p, a * b * 4).tobytes()).hexdigest() for p, (a, b) in zip(bufs, sizes)]
        print("cfg4 mips %s: %.4f ms per chain, %s" % (name, best * 1e3, "IDENTICAL to the reference golden" if got == GOLD["cfg4_" + name]["levels"] else "DIFFERS from the reference golden"))
    bc3 = [dmalloc(dx.compute_pitch(77, a, b)[1]) for a, b in sizes]
    dsts = [dx.capi.device_image(p, a, b, 77) for p, (a, b) in zip(bc3, sizes)]
    f = lambda: ctx.compress_many_device(levels, dsts, 0, 0.5)
    f(); ctx.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(5): f()
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / 5)
    got = [hashlib.sha256(d2h(p, dx.compute_pitch(77, a, b)[1]).tobytes()).hexdigest() for p, (a, b) in zip(bc3, sizes)]
    print("cfg4 BC3 of the box chain: %.4f ms, %s" % (best * 1e3, "IDENTICAL to the reference golden" if got == GOLD["cfg4_box"]["bc3_levels"] else "DIFFERS from the reference golden"))
ctx.close()
