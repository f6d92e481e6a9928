#!/usr/bin/env python3
"""DEVELOPMENT TOOL: pruned (default) vs unpruned (DXTEX_BC7_NO_PRUNE=1) BC7 payloads must be identical.
usage: python tools/bc7_prune_check.py            -> runs itself twice and compares digests
       python tools/bc7_prune_check.py --emit     -> prints one digest line per case"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)


def emit():
    import numpy as np, torch
    import directxtex_amd as dx
    if "--dev" in sys.argv: dx.capi.load(dev=True)
    from directxtex_amd import synth
    import bench
    ctx = dx.Context(0); dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    yy, xx = np.mgrid[0:2048, 0:2048]
    smooth = np.stack([(xx / 8) % 256, (yy / 8) % 256, ((xx + yy) / 16) % 256, 255 - (xx / 16) % 256], -1).astype(np.uint8)
    cases = [("bench 4096 opaque", bench.make_image(0), 0),
             ("noise+alpha 2048", np.tile(synth.rgba8(1024, 1024, seed=9, alpha="smooth"), (2, 2, 1)), 0),
             ("random alpha 1024", synth.rgba8(1024, 1024, seed=11, alpha="random"), 0),
             ("gradients 2048", smooth, 0),
             ("3subsets 1024", synth.rgba8(1024, 1024, seed=12, alpha="opaque"), 0x80000),
             ("pure noise 1024", np.random.default_rng(1).integers(0, 256, (1024, 1024, 4), dtype=np.uint8), 0)]
    if "--bc6h" in sys.argv:
        rng = np.random.default_rng(3)
        hdr = lambda im, s: (im.astype(np.float32) / 255.0 * s).astype(np.float16)
        spiky = (np.exp2(rng.uniform(-8, 6, (1024, 1024, 1))) * (synth.rgba8(1024, 1024, seed=4).astype(np.float32) / 255.0)).astype(np.float16)
        neg = (synth.rgba8(1024, 1024, seed=6).astype(np.float32) / 255.0 * 4.0 - 2.0).astype(np.float16)
        cases = [("hdr bench-like 4096", hdr(bench.make_image(0), 8.0), 95), ("hdr smooth 2048", hdr(smooth, 2.0), 95),
                 ("hdr spiky 1024", spiky, 95), ("hdr signed 1024", neg, 96), ("hdr noise 1024", hdr(cases[5][1], 16.0), 95)]
    for name, img, flags in cases:
        h, w = img.shape[:2]
        src = torch.from_numpy(np.ascontiguousarray(img).view(np.uint8)).to(dev)
        sfmt, dfmt = (10, flags) if img.dtype == np.float16 else (28, 98)
        if img.dtype == np.float16: flags = 0
        out = torch.empty(dx.compute_pitch(dfmt, w, h)[1], dtype=torch.uint8, device=dev)
        run = lambda: ctx.compress_device(src.data_ptr(), w, h, sfmt, out.data_ptr(), dfmt, flags, 0.5)
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%s|%s|%.1f ms %.1f Mtexels/s" % (name, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16], dt * 1e3, w * h / dt / 1e6), flush=True)


if "--emit" in sys.argv:
    emit()
else:
    runs = []
    extra = ["--bc6h"] if "--bc6h" in sys.argv else []
    envs = [("unpruned", {"DXTEX_BC7_NO_PRUNE": "1", "DXTEX_BC6H_NO_PRUNE": "1"}), ("default", {})] + [("order " + o, {"DXTEX_BC7_ORDER": o}) for o in sys.argv[1:] if o[0].isdigit()]
    for tag, env in envs:
        r = subprocess.run([sys.executable, __file__, "--emit", "--dev"] + extra, env=dict(os.environ, **env), capture_output=True, text=True)
        runs.append([l.split("|") for l in r.stdout.splitlines() if l.count("|") == 2])
        if r.returncode != 0: print(r.stderr[-2000:])
    ok = all(len(r) == len(runs[0]) for r in runs) and bool(runs[0])
    for i, base in enumerate(runs[0]):
        line = "%-20s unpruned %-28s" % (base[0], base[2])
        for (tag, _), r in zip(envs[1:], runs[1:]):
            same = i < len(r) and r[i][1] == base[1]; ok &= same
            line += " | %s: %s %s" % (tag, "same" if same else "DIFFERENT", r[i][2] if i < len(r) else "?")
        print(line)
    print("ALL IDENTICAL" if ok else "MISMATCH")
