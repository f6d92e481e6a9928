#!/usr/bin/env python3
"""DEVELOPMENT TOOL: per-kernel times of one 4096^2 BC7 encode (the context's hipEvent marks) and wall time of the small kernels, for
A/B runs of development knobs: `DXTEX_...=... python tools/r03_probe.py --dev [bc7] [convert] [bc15] [decode]`."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import directxtex_amd as dx
if "--dev" in sys.argv:
    sys.argv.remove("--dev"); dx.capi.load(dev=True)      # the -DDXTEX_DEV build: the only one that reads DXTEX_* knobs
from directxtex_amd import synth

what = set(sys.argv[1:]) or {"bc7"}
ctx = dx.Context(0); dev = torch.device("cuda", 0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
W = H = int(os.environ.get("PROBE_SIZE", "4096"))
img = synth.survey_rgba8(W, H, 2, "opaque")
src = torch.from_numpy(img).to(dev)


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e3


if "bc7" in what:
    dst = torch.empty(dx.compute_pitch(98, W, H)[1], dtype=torch.uint8, device=dev)
    f = lambda: ctx.compress_device(src.data_ptr(), W, H, 28, dst.data_ptr(), 98, 0, 0.5)
    print("bc7 %d^2: %.3f ms per image (wall, 3 images)" % (W, timed(f, 3)))
    ctx.profile_begin(); f(); k = ctx.profile_end()
    tot = sum(ms for ms, n in k.values())
    print("  serial kernel sum %.2f ms" % tot)
    for name, (ms, n) in sorted(k.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("PROBE_TOP", "14"))]:
        print("  %-40s %8.3f ms x%d" % (name, ms / n, n))
    import hashlib
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json")))["cases"]["cfg2_bc7_4096"]["sha256"]
    sha = hashlib.sha256(dst.cpu().numpy().tobytes()).hexdigest()
    print("  payload sha256", sha[:16], ("IDENTICAL to the reference golden" if sha == gold else "DIFFERS from the reference golden") if W == 4096 else "")
if "bc6h" in what:
    import hashlib, json
    hdr = torch.from_numpy(synth.survey_rgba16f(W, H, 3)).to(dev)
    dst6 = torch.empty(dx.compute_pitch(95, W, H)[1], dtype=torch.uint8, device=dev)
    f6 = lambda: ctx.compress_device(hdr.data_ptr(), W, H, 10, dst6.data_ptr(), 95, 0, 0.5)
    print("bc6h 4096^2: %.3f ms per image (wall, 2 images)" % timed(f6, 2))
    ctx.profile_begin(); f6(); k = ctx.profile_end()
    for name, (ms, n) in sorted(k.items(), key=lambda kv: kv[0]):
        print("  %-40s %8.3f ms x%d" % (name, ms, n))
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json")))["cases"]["cfg3_bc6h_uf16_4096"]["sha256"]
    sha = hashlib.sha256(dst6.cpu().numpy().tobytes()).hexdigest()
    print("  payload", "IDENTICAL to the reference golden" if sha == gold else "DIFFERS from the reference golden")
    del hdr, dst6
if "convert" in what:
    for sf, df, sb, db in ((28, 10, 4, 8), (10, 28, 8, 4), (28, 2, 4, 16), (28, 87, 4, 4), (2, 10, 16, 8)):
        s = torch.zeros(W * H * sb, dtype=torch.uint8, device=dev); s[:W * H * 4] = src.reshape(-1)[:W * H * 4]
        d = torch.empty(W * H * db, dtype=torch.uint8, device=dev)
        si = dx.capi.device_image(s.data_ptr(), W, H, sf); di = dx.capi.device_image(d.data_ptr(), W, H, df)
        ms = timed(lambda: ctx._check(dx.capi._lib.dxtex_convert_device(ctx._h, ctypes.byref(si), ctypes.byref(di), 0, 0.5), "convert"))
        print("convert %3d -> %3d 4096^2: %.4f ms = %.2f TB/s algorithmic" % (sf, df, ms, W * H * (sb + db) / ms / 1e9))
if "bc15" in what:
    for fmt, bpt in ((71, 4.5), (74, 5), (77, 5), (80, 4.5), (83, 5)):
        dst = torch.empty(dx.compute_pitch(fmt, W, H)[1], dtype=torch.uint8, device=dev)
        ms = timed(lambda: ctx.compress_device(src.data_ptr(), W, H, 28, dst.data_ptr(), fmt, 0, 0.5))
        print("bc format %d 4096^2: %.4f ms = %.2f TB/s algorithmic" % (fmt, ms, W * H * bpt / ms / 1e9))
if "decode" in what:
    back = torch.empty(W * H * 4, dtype=torch.uint8, device=dev)
    for fmt in (71, 77, 83, 98):
        dst = torch.empty(dx.compute_pitch(fmt, W, H)[1], dtype=torch.uint8, device=dev)
        ctx.compress_device(src.data_ptr(), W, H, 28, dst.data_ptr(), fmt, dx.TEX_COMPRESS_BC7_QUICK if fmt == 98 else 0, 0.5)
        ms = timed(lambda: ctx.decompress_device(dst.data_ptr(), W, H, fmt, back.data_ptr(), 28))
        print("decode %d -> RGBA8 4096^2: %.4f ms = %.2f TB/s algorithmic" % (fmt, ms, (dst.numel() + W * H * 4) / ms / 1e9))
    for fmt, tgt, bpt, bb in ((80, 61, 1, 0.5), (83, 49, 2, 1.0)):                      # BC4 -> R8, BC5 -> R8G8: the default targets
        dst = torch.empty(dx.compute_pitch(fmt, W, H)[1], dtype=torch.uint8, device=dev)
        ctx.compress_device(src.data_ptr(), W, H, 28, dst.data_ptr(), fmt, 0, 0.5)
        ms = timed(lambda: ctx.decompress_device(dst.data_ptr(), W, H, fmt, back.data_ptr(), tgt))
        print("decode %d -> %d 4096^2: %.4f ms = %.2f TB/s algorithmic" % (fmt, tgt, ms, W * H * (bpt + bb) / ms / 1e9))
    # every BC7 mode / partition / rotation (arbitrary blocks), and BC6H of the encoder's output to RGBA16F
    rnd = torch.randint(0, 256, (W * H,), dtype=torch.uint8, device=dev)
    ms = timed(lambda: ctx.decompress_device(rnd.data_ptr(), W, H, 98, back.data_ptr(), 28))
    print("decode 98 (arbitrary blocks) -> RGBA8 4096^2: %.4f ms = %.2f TB/s algorithmic" % (ms, (W * H * 5) / ms / 1e9))
    srch = (src.view(H, W, 4).float() / 255.0).half().contiguous()
    b6 = torch.empty(W * H, dtype=torch.uint8, device=dev); back16 = torch.empty(W * H * 8, dtype=torch.uint8, device=dev)
    ctx.compress_device(srch.data_ptr(), W, H, 10, b6.data_ptr(), 95, 0, 0.5)
    ms = timed(lambda: ctx.decompress_device(b6.data_ptr(), W, H, 95, back16.data_ptr(), 10))
    print("decode 95 -> RGBA16F 4096^2: %.4f ms = %.2f TB/s algorithmic" % (ms, (W * H * 9) / ms / 1e9))
    ms = timed(lambda: ctx.decompress_device(rnd.data_ptr(), W, H, 95, back16.data_ptr(), 10))
    print("decode 95 (arbitrary blocks) -> RGBA16F 4096^2: %.4f ms = %.2f TB/s algorithmic" % (ms, (W * H * 9) / ms / 1e9))
if "mips" in what:
    big = torch.from_numpy(synth.survey_rgba8(8192, 8192, 4, "random")).to(dev)
    sizes = []; w = h = 8192
    while True:
        sizes.append((w, h))
        if w == 1 and h == 1: break
        w, h = max(1, w >> 1), max(1, h >> 1)
    bufs = [big.reshape(-1)] + [torch.empty(a * b * 4, dtype=torch.uint8, device=dev) for a, b in sizes[1:]]
    levels = [dx.capi.device_image(t.data_ptr(), a, b, 28) for t, (a, b) in zip(bufs, sizes)]
    import hashlib, json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json")))["cases"]
    for name, flt in (("box", dx.TEX_FILTER_BOX), ("cubic", dx.TEX_FILTER_CUBIC)):
        ms = timed(lambda: ctx.generate_mips_device(levels, flt), 10)
        same = [hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest() for t in bufs] == gold["cfg4_" + name]["levels"]
        print("mip chain 8192^2 %s: %.4f ms = %.2f TB/s algorithmic, %s" % (name, ms, 447392420 / ms / 1e9, "IDENTICAL to golden" if same else "DIFFERS"))
    ctx.generate_mips_device(levels, dx.TEX_FILTER_BOX)
    bc3 = [torch.empty(dx.compute_pitch(77, a, b)[1], dtype=torch.uint8, device=dev) for a, b in sizes]
    dsts = [dx.capi.device_image(t.data_ptr(), a, b, 77) for t, (a, b) in zip(bc3, sizes)]
    ms = timed(lambda: ctx.compress_many_device(levels, dsts, 0, 0.5), 10)
    same = [hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest() for t in bc3] == gold["cfg4_box"]["bc3_levels"]
    print("BC3 of the 14 levels: %.4f ms, %s" % (ms, "IDENTICAL to golden" if same else "DIFFERS"))
ctx.close()
