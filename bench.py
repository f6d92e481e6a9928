#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X DirectXTex hot path.

Metric (BASELINE.json): Mtexels/s of BC7 encode, 4096x4096 RGBA8, TEX_COMPRESS_DEFAULT (cfg2).

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of DirectX::Compress' hot path (dxtex_compress_device: source resident in HBM, BC7 payload written to
HBM) over one 4096^2 synthetic image per GPU: SURVEY.md section 8d's cfg2 recipe (directxtex_amd.synth.survey_rgba8, LCG seed
2 + rank). N > 1 is launched by the driver through torch.distributed.run, one rank per GPU; images are sharded one-per-GPU (the
path has no exchange step, so there is no data-path collective: the only collectives are the timing barrier and the MAX over
ranks). Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     - dominant kernel: algorithmic bytes per launch / its mean duration (hipEvents on the launch stream, recorded in
                 a separate, untimed profiling pass through dxtex_ctx_profile_*), against 8 TB/s HBM. BC7 at the reference's
                 search depth is VALU-bound, so the HBM fraction is small by design; `valu` gives the kernel's VALU issue
                 utilisation against the measured gfx950 issue rates (profiles/r02_valu_rates.md), `all_kernels_ms` every kernel.
                 The same evidence as flat scalars (for consumers that keep scalars only): valu_issue_utilisation,
                 active_lanes_per_valu_inst, lds_bank_conflict_share, traffic_whole_step[_over_algorithmic] (every launch of one
                 image), late_phase_kernel_ms.
  parity       - the payload of the timed run against the committed digests of the reference's output for the same image
                 (tests/golden/fullsize.json: whole 4096^2 image, 1 048 576 blocks) and, live, against the reference run on
                 this box's host cores on a bounded sample of block rows; PSNR of the whole image (decoded on the GPU).
  cpu_baseline - the reference's own encoder (oracle/_ref: DirectX::Compress with TEX_COMPRESS_PARALLEL = CompressBC_Parallel,
                 OpenMP over blocks) timed on this box's host cores on that bounded sample (rank 0, N = 1 only; --cpu-full runs
                 the whole image, ~3.5 min on 128 threads). Reported, not a target.
  other_workloads - cfg3, cfg4, a cfg5 shard and the other codecs, each with its own roofline object; for N > 1 one image split over
                 the ranks (bc7_4096_split) and, where one process sees several GPUs, over contexts of this process
                 (bc7_4096_inprocess_split: dxtex_compress_multi). tools/perf_guard.py compares their times with the last committed line.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH = HEIGHT = 4096
ALGO_BYTES_PER_TEXEL = 5.0          # 4 B read + 1 B written per texel for RGBA8 -> BC7
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s
PCIE_PEAK_GBS = 63.0                # PCIe 5.0 x16, one direction (MI355X_MICROARCH.md)
BAND_ROWS = 16                      # block rows per digest band (tests/golden/make_golden_fullsize.py)
TEX_COMPRESS_PARALLEL = 0x10000000


def make_image(rank):
    """4096^2 RGBA8, LCG seed 2 (+ rank: every GPU compresses its own image), opaque."""
    from directxtex_amd import synth
    return synth.survey_rgba8(WIDTH, HEIGHT, 2 + rank, "opaque")


def hbm_roofline(algo_bytes, ms, kernel=None, traffic=None, time_key="kernel_ms"):
    """time_key names what `ms` is: "kernel_ms" for device time of resident work, "wall_ms" where host copies and PCIe are inside."""
    achieved = algo_bytes / (ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
         "traffic": traffic, "algorithmic_bytes_per_launch": int(algo_bytes), time_key: round(ms, 4)}
    if kernel:
        r["kernel"] = kernel
    return r


def band_sha(payload, width, height):
    nbw, nbh = (width + 3) // 4, (height + 3) // 4
    rows = np.ascontiguousarray(payload, np.uint8).reshape(nbh, nbw * 16)
    return [hashlib.sha256(rows[r:r + BAND_ROWS].tobytes()).hexdigest() for r in range(0, nbh, BAND_ROWS)]


def cpu_baseline(img, payload, full, budget_s=20.0):
    """The reference on the host cores: on bands of 16 block rows of the benchmark image spread over its height (a band is an image of
    its own for a block codec), sized for ~budget_s of wall time, or on the whole image. Returns (cpu_baseline object, parity dict)."""
    import oracle
    if not oracle.have_ref():
        return None, {}
    threads = oracle.ref_num_threads()
    rows = BAND_ROWS * 4
    nbands = HEIGHT // rows
    got = np.ascontiguousarray(payload, np.uint8).reshape(HEIGHT // 4, (WIDTH // 4) * 16)
    texels = 0; secs = 0.0; same = 0; blocks = 0
    done = []; band_secs = {}

    def run_band(b):
        nonlocal texels, secs, same, blocks
        crop = np.ascontiguousarray(img[b * rows:(b + 1) * rows])
        t0 = time.perf_counter()
        ref = oracle.ref_compress_image(crop, WIDTH, crop.shape[0], 28, 98, TEX_COMPRESS_PARALLEL, 0.5)
        band_secs[b] = time.perf_counter() - t0
        secs += band_secs[b]
        texels += crop.shape[0] * WIDTH
        g = got[b * BAND_ROWS:(b + 1) * BAND_ROWS].reshape(-1, 16)
        same += int((g == ref.reshape(-1, 16)).all(axis=1).sum()); blocks += g.shape[0]
        done.append(b)

    if full:
        for b in range(nbands):
            run_band(b)
    else:
        # the first band calibrates: then as many more, spread over the image's height, as fit the budget
        run_band(nbands // 2)
        n = int(max(0, min(nbands - 1, budget_s / max(secs, 1e-3) - 1)))
        for i in range(n):
            b = int(round(i * (nbands - 1) / max(1, n - 1))) if n > 1 else 0
            if b not in done:
                run_band(b)
    picks = done
    sample = "the whole 4096x4096 image" if full else f"{len(picks)} bands of {rows} rows x {WIDTH} ({texels} texels) spread over the benchmark image"
    raw = texels / secs / 1e6
    cpu = {"value": round(raw, 5), "unit": "Mtexels/s", "cores": threads, "kind": "reference",
           "sample": f"{sample}; DirectX::Compress -> CompressBC_Parallel (OpenMP over blocks), D3DXEncodeBC7 flags=0, {secs:.1f} s"}
    if full:
        cpu["band_seconds"] = [round(band_secs[b], 4) for b in range(nbands)]
    else:
        # The bands are not equally expensive (flat blocks end at mode 6, noisy ones search every mode): weight the sample with the
        # per-band cost of a whole-image run of the reference on this image (profiles/r06_cpu_bands.json, from `bench.py --cpu-full`
        # on the GPU box), so that `value` estimates the WHOLE image's rate: seconds(whole) ~ seconds(sample) * cost(all) / cost(sample).
        table = band_cost_table(nbands)
        if table:
            frac = sum(table[b] for b in picks) / sum(table)
            est = WIDTH * HEIGHT / (secs / frac) / 1e6
            cpu["value"] = round(est, 5)
            cpu["raw_sample_value"] = round(raw, 5)
            cpu["weighting"] = (f"sample seconds scaled by the sample's share of the whole image's reference time ({frac:.4f} of it in "
                                f"{len(picks)}/{nbands} of the bands; profiles/r06_cpu_bands.json): an estimate of the whole-image rate")
    return cpu, {"live_reference_blocks_compared": blocks, "live_reference_blocks_identical": same}


def band_cost_table(nbands):
    """Per-band seconds of the reference over the whole benchmark image (one `--cpu-full` run on the GPU box, committed)."""
    fp = os.path.join(ROOT, "profiles", "r06_cpu_bands.json")
    if not os.path.exists(fp):
        return None
    t = json.load(open(fp)).get("band_seconds")
    return t if t and len(t) == nbands and min(t) > 0 else None


def gpu_psnr(ctx, dev, src, dst, fmt_src, fmt_bc, width, height):
    """RGB PSNR (texdiag's formula) of a BC payload against its source, decoded and reduced on the GPU."""
    import torch
    back = torch.empty(width * height * 4, dtype=torch.uint8, device=dev)
    ctx.decompress_device(dst.data_ptr(), width, height, fmt_bc, back.data_ptr(), 28)
    mse = ctx.compute_mse_device(back.data_ptr(), 28, src.data_ptr(), fmt_src, width, height)
    return float(10.0 * np.log10(3.0 / max(1e-30, float(mse[0] + mse[1] + mse[2]))))


def resident_end_to_end(ctx, dev, run, src_np, out_bytes, check=None, reps=3):
    """SURVEY 8d's second number: host buffer -> H2D -> kernels -> D2H -> host buffer, as ONE stream-ordered submission on the context
    (dxtex_memcpy_h2d_async, the *_device steps, dxtex_memcpy_d2h_async, one synchronize): the source crosses PCIe once, the final
    payload once, intermediates stay in HBM. `run(src_ptr, dst_ptr)` queues the steps. Measured from pageable numpy memory and from
    page-locked memory (dxtex_host_alloc); bytes moved come from the context's own transfer counters."""
    import torch
    src_np = np.ascontiguousarray(src_np).reshape(-1).view(np.uint8)
    d_src = torch.empty(src_np.nbytes, dtype=torch.uint8, device=dev)
    d_dst = torch.empty(out_bytes, dtype=torch.uint8, device=dev)
    pin_src = ctx.host_alloc(src_np.nbytes); pin_src[:] = src_np
    pin_dst = ctx.host_alloc(out_bytes)
    out = {}
    try:
        for name, hs, hd in (("pageable", src_np, np.zeros(out_bytes, np.uint8)), ("pinned", pin_src, pin_dst)):
            best = None
            for k in range(reps + 1):                        # the first pass warms allocations inside the steps
                hd[:] = 0
                ctx.transfer_bytes(reset=True)
                t0 = time.perf_counter()
                ctx.upload(d_src.data_ptr(), hs)
                run(d_src.data_ptr(), d_dst.data_ptr())
                ctx.download(hd, d_dst.data_ptr())
                ctx.synchronize()
                dt = time.perf_counter() - t0
                if k > 0:
                    best = dt if best is None else min(best, dt)
            up, down = ctx.transfer_bytes()
            e = {"ms": round(best * 1e3, 3), "h2d_bytes": up, "d2h_bytes": down,
                 "pcie_GBs": round((up + down) / best / 1e9, 2), "pcie_frac_of_63GBs": round((up + down) / best / 1e9 / PCIE_PEAK_GBS, 3)}
            if check is not None:
                e["identical_to_reference_golden"] = bool(check(hd))
            out[name] = e
        out["transfer_bytes_over_source_plus_payload"] = round((up + down) / float(src_np.nbytes + out_bytes), 4)
    finally:
        ctx.host_free(pin_src.ctypes.data); ctx.host_free(pin_dst.ctypes.data)
    return out


def other_workloads(ctx, dev, img, rank, world):
    """The other configurations of BASELINE.json, measured AFTER the timed region (reported, not the metric): device-resident
    inputs, per-call wall time with a stream sync, algorithmic bytes per SURVEY.md section 8d against the 8 TB/s HBM roofline."""
    import torch
    import directxtex_amd as dx
    from directxtex_amd import synth
    RGBA8, RGBA16F = dx.DXGI_FORMAT_R8G8B8A8_UNORM, dx.DXGI_FORMAT_R16G16B16A16_FLOAT
    out = {}

    def timed(fn, n):
        # best of three batches: the sub-millisecond kernels are timed through the host's launch path, which a busy host (the CPU
        # baseline's OpenMP threads winding down, another rank's synthesis) can hold up for a batch
        fn(); torch.cuda.synchronize(dev)
        best = None
        for _ in range(3 if n >= 5 else 1):
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / n
            best = dt if best is None else min(best, dt)
        return best

    def entry(dt, texels, algo_bytes, profile=None, **kw):
        e = {"ms": round(dt * 1e3, 3), "Mtexels_s": round(texels / dt / 1e6, 1), "roofline": hbm_roofline(algo_bytes, dt * 1e3)}
        if profile:
            e["profile"] = profile
        e.update(kw)
        return e

    tex = WIDTH * HEIGHT
    src = torch.from_numpy(img).to(dev)
    for name, fmt, bpt, n in (("bc1", dx.DXGI_FORMAT_BC1_UNORM, 4.5, 20), ("bc3", dx.DXGI_FORMAT_BC3_UNORM, 5.0, 20), ("bc5", dx.DXGI_FORMAT_BC5_UNORM, 5.0, 20)):
        dst = torch.empty(dx.compute_pitch(fmt, WIDTH, HEIGHT)[1], dtype=torch.uint8, device=dev)
        dt = timed(lambda: ctx.compress_device(src.data_ptr(), WIDTH, HEIGHT, RGBA8, dst.data_ptr(), fmt, 0, 0.5), n)
        out[f"{name}_4096"] = entry(dt, tex, tex * bpt, "profiles/r04_kernels.md")
        if name != "bc5":
            nb = dst.numel()
            out[f"{name}_4096"]["end_to_end"] = resident_end_to_end(
                ctx, dev, lambda s, d, fmt=fmt: ctx.compress_device(s, WIDTH, HEIGHT, RGBA8, d, fmt, 0, 0.5), img, nb)
    # cfg1: one 256 x 256 RGBA8 image -> BC1 (BASELINE.json configs[0]; SURVEY 8d seed-1 recipe), kernel-only and end to end
    c1 = synth.rgba8(256, 256, seed=1, alpha="opaque")
    c1d = torch.from_numpy(c1).to(dev)
    c1o = torch.empty(dx.compute_pitch(dx.DXGI_FORMAT_BC1_UNORM, 256, 256)[1], dtype=torch.uint8, device=dev)
    dt = timed(lambda: ctx.compress_device(c1d.data_ptr(), 256, 256, RGBA8, c1o.data_ptr(), dx.DXGI_FORMAT_BC1_UNORM, 0, 0.5), 50)
    e = entry(dt, 256 * 256, 256 * 256 * 4.5)
    e["end_to_end"] = resident_end_to_end(ctx, dev, lambda s, d: ctx.compress_device(s, 256, 256, RGBA8, d, dx.DXGI_FORMAT_BC1_UNORM, 0, 0.5), c1, c1o.numel(), reps=10)
    out["cfg1_bc1_256"] = e
    # the reference's faster / slower BC7 settings on the same image (TEX_COMPRESS_BC7_QUICK: mode 6 only; BC7_USE_3SUBSETS: + modes 0, 2)
    dst = torch.empty(dx.compute_pitch(dx.DXGI_FORMAT_BC7_UNORM, WIDTH, HEIGHT)[1], dtype=torch.uint8, device=dev)
    for name, fl, n in (("bc7_quick_4096", dx.TEX_COMPRESS_BC7_QUICK, 5), ("bc7_3subsets_4096", 0x80000, 1)):
        dt = timed(lambda: ctx.compress_device(src.data_ptr(), WIDTH, HEIGHT, RGBA8, dst.data_ptr(), dx.DXGI_FORMAT_BC7_UNORM, fl, 0.5), n)
        out[name] = entry(dt, tex, tex * 5.0)
    # the headline codec on other content (the rate depends on what can be pruned): cfg2's variant with random alpha, and round 1's
    # benchmark image (hash-noise recipe synth.rgba8, four 1024^2 tiles), kept for continuity with BENCH_r01
    alt = torch.from_numpy(synth.survey_rgba8(WIDTH, HEIGHT, 2, "random")).to(dev)
    dt = timed(lambda: ctx.compress_device(alt.data_ptr(), WIDTH, HEIGHT, RGBA8, dst.data_ptr(), dx.DXGI_FORMAT_BC7_UNORM, 0, 0.5), 1)
    out["bc7_4096_cfg2_random_alpha"] = entry(dt, tex, tex * 5.0)
    tiles = [synth.rgba8(1024, 1024, seed=32 + i, alpha="opaque") for i in range(4)]
    r01 = np.ascontiguousarray(np.concatenate([np.concatenate([tiles[(x + y) % 4] for x in range(4)], axis=1) for y in range(4)], axis=0))
    alt = torch.from_numpy(r01).to(dev)
    dt = timed(lambda: ctx.compress_device(alt.data_ptr(), WIDTH, HEIGHT, RGBA8, dst.data_ptr(), dx.DXGI_FORMAT_BC7_UNORM, 0, 0.5), 2)
    out["bc7_4096_round1_image"] = entry(dt, tex, tex * 5.0, note="BENCH_r01's image: 86.2 Mtexels/s in round 1")
    del alt
    # cfg3: 4096^2 RGBA16F -> BC6H_UF16 (SURVEY 8d seed-3 recipe), 9 B/texel
    hdr_np = synth.survey_rgba16f(WIDTH, HEIGHT, 3)
    hdr = torch.from_numpy(hdr_np).to(dev)
    dst6 = torch.empty(dx.compute_pitch(dx.DXGI_FORMAT_BC6H_UF16, WIDTH, HEIGHT)[1], dtype=torch.uint8, device=dev)
    dt = timed(lambda: ctx.compress_device(hdr.data_ptr(), WIDTH, HEIGHT, RGBA16F, dst6.data_ptr(), dx.DXGI_FORMAT_BC6H_UF16, 0, 0.5), 2)
    e = entry(dt, tex, tex * 9.0, "profiles/r04_kernels.md")
    gold = golden_case("cfg3_bc6h_uf16_4096")
    if gold:
        e["identical_to_reference_golden"] = band_sha(dst6.cpu().numpy(), WIDTH, HEIGHT) == gold["bands"]
    e["end_to_end"] = resident_end_to_end(
        ctx, dev, lambda s, d: ctx.compress_device(s, WIDTH, HEIGHT, RGBA16F, d, dx.DXGI_FORMAT_BC6H_UF16, 0, 0.5), hdr_np, dst6.numel(),
        check=(lambda p: band_sha(p, WIDTH, HEIGHT) == gold["bands"]) if gold else None, reps=2)
    out["cfg3_bc6h_uf16_4096"] = e
    del hdr, hdr_np
    # decoders, 4096^2 to their default targets (block bytes read + target bytes written per texel): BC7 of the headline's own payload
    # (all modes), BC1 / BC3 of this image, BC6H of cfg3's payload
    back = torch.empty(tex * 8, dtype=torch.uint8, device=dev)
    ctx.compress_device(src.data_ptr(), WIDTH, HEIGHT, RGBA8, dst.data_ptr(), dx.DXGI_FORMAT_BC7_UNORM, 0, 0.5)
    dt = timed(lambda: ctx.decompress_device(dst.data_ptr(), WIDTH, HEIGHT, dx.DXGI_FORMAT_BC7_UNORM, back.data_ptr(), RGBA8), 20)
    out["bc7_decode_4096"] = entry(dt, tex, tex * 5.0, "profiles/r04_kernels.md")
    for name, fmt, bpt in (("bc1", dx.DXGI_FORMAT_BC1_UNORM, 4.5), ("bc3", dx.DXGI_FORMAT_BC3_UNORM, 5.0)):
        ctx.compress_device(src.data_ptr(), WIDTH, HEIGHT, RGBA8, dst.data_ptr(), fmt, 0, 0.5)
        dt = timed(lambda: ctx.decompress_device(dst.data_ptr(), WIDTH, HEIGHT, fmt, back.data_ptr(), RGBA8), 20)
        out[f"{name}_decode_4096"] = entry(dt, tex, tex * bpt, "profiles/r04_kernels.md")
    dt = timed(lambda: ctx.decompress_device(dst6.data_ptr(), WIDTH, HEIGHT, dx.DXGI_FORMAT_BC6H_UF16, back.data_ptr(), RGBA16F), 20)
    out["bc6h_decode_4096"] = entry(dt, tex, tex * 9.0, "profiles/r04_kernels.md")
    del back, dst6
    # cfg4: 8192^2 RGBA8 (seed 4, random alpha) full mip chain (box, cubic), then BC3 of all 14 levels
    big_np = synth.survey_rgba8(8192, 8192, 4, "random")
    big = torch.from_numpy(big_np).to(dev)
    w = h = 8192
    sizes = []
    while True:
        sizes.append((w, h))
        if w == 1 and h == 1:
            break
        w, h = max(1, w >> 1), max(1, h >> 1)
    bufs = [big.reshape(-1)] + [torch.empty(a * b * 4, dtype=torch.uint8, device=dev) for a, b in sizes[1:]]
    levels = [dx.capi.device_image(t.data_ptr(), a, b, RGBA8) for t, (a, b) in zip(bufs, sizes)]
    chain_bytes = sum(a * b * 4 for a, b in sizes[:-1]) + sum(a * b * 4 for a, b in sizes[1:])          # 447 392 420 B
    chain_tex = sum(a * b for a, b in sizes[1:])
    for name, flt in (("box", dx.TEX_FILTER_BOX), ("cubic", dx.TEX_FILTER_CUBIC)):
        dt = timed(lambda: ctx.generate_mips_device(levels, flt), 5)
        e = entry(dt, chain_tex, chain_bytes, "profiles/r04_kernels.md")
        gold = golden_case(f"cfg4_{name}")
        if gold:
            e["identical_to_reference_golden"] = [hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest() for t in bufs] == gold["levels"]
        out[f"cfg4_mips_{name}_8192"] = e
    ctx.generate_mips_device(levels, dx.TEX_FILTER_BOX)
    bc3 = [torch.empty(dx.compute_pitch(dx.DXGI_FORMAT_BC3_UNORM, a, b)[1], dtype=torch.uint8, device=dev) for a, b in sizes]
    dsts = [dx.capi.device_image(t.data_ptr(), a, b, dx.DXGI_FORMAT_BC3_UNORM) for t, (a, b) in zip(bc3, sizes)]
    dt = timed(lambda: ctx.compress_many_device(levels, dsts, 0, 0.5), 5)
    tex4 = sum(a * b for a, b in sizes)
    e = entry(dt, tex4, 447392452, "profiles/r04_kernels.md")
    gold = golden_case("cfg4_box")
    if gold:
        e["identical_to_reference_golden"] = [hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest() for t in bc3] == gold["bc3_levels"]
    out["cfg4_mipchain_bc3_8192"] = e
    # cfg4 end to end, resident: the 8192^2 source up once (256 MiB), GenerateMipMaps (box) and Compress -> BC3 of all 14 levels on the device,
    # the BC3 chain down once (85 MiB): texconv's mipmaps -> compress steps (texconv.cpp:3434, 3711) without the round trips between them
    offs = np.concatenate([[0], np.cumsum([t.numel() for t in bc3])]).astype(np.int64)
    lv_off = np.concatenate([[0], np.cumsum([a * b * 4 for a, b in sizes[1:]])]).astype(np.int64)
    mips_d = torch.empty(int(lv_off[-1]), dtype=torch.uint8, device=dev)

    def cfg4_run(s_ptr, d_ptr):
        lv = [dx.capi.device_image(s_ptr, 8192, 8192, RGBA8)] + [dx.capi.device_image(mips_d.data_ptr() + int(lv_off[i]), a, b, RGBA8) for i, (a, b) in enumerate(sizes[1:])]
        ds = [dx.capi.device_image(d_ptr + int(offs[i]), a, b, dx.DXGI_FORMAT_BC3_UNORM) for i, (a, b) in enumerate(sizes)]
        ctx.generate_mips_device(lv, dx.TEX_FILTER_BOX)
        ctx.compress_many_device(lv, ds, 0, 0.5)

    def cfg4_check(p):
        return [hashlib.sha256(p[int(offs[i]):int(offs[i + 1])].tobytes()).hexdigest() for i in range(len(sizes))] == gold["bc3_levels"]

    out["cfg4_end_to_end"] = resident_end_to_end(ctx, dev, cfg4_run, big_np, int(offs[-1]), check=cfg4_check if gold else None, reps=2)
    out["cfg4_end_to_end"]["workload"] = ("cfg4: 8192^2 RGBA8 host image -> H2D -> GenerateMipMaps(box, 14 levels) -> Compress(BC3, all levels) -> D2H of the BC3 chain, "
                                          "one stream-ordered submission")
    out["cfg4_end_to_end"]["kernels_only_ms"] = round(out["cfg4_mips_box_8192"]["ms"] + out["cfg4_mipchain_bc3_8192"]["ms"], 3)
    del big, bufs, bc3, mips_d, big_np
    return out


CFG5_SIDE = 2048
CFG5_CHUNK_TEXELS = 32 << 20         # dxtex_compress_many's chunk size (csrc/capi.cpp): eight 2048^2 images


def kernel_sources_sha256():
    """Digest of the sources of the BC7 search kernels: profiles/pmc_traffic.json is stamped with it, so counters measured on another
    build are never republished (see main)."""
    h = hashlib.sha256()
    for name in ("bc7_encode.hip", "bc7_core.h", "search_common.h"):
        h.update(open(os.path.join(ROOT, "directxtex_amd", "csrc", name), "rb").read())
    return h.hexdigest()


def cfg5_shard(ctx, dev, rank, world, per_rank):
    """cfg5 (1024 x 2048^2 RGBA8 -> BC7, image i on GPU i mod N): this rank's first `per_rank` images of its shard, host pointers in,
    host pointers out through dxtex_compress_many (pinned double-buffered H2D / D2H overlapped with the search kernels). 16 distinct
    host images are cycled. One untimed pass first (one-time allocations), then the timed one. Every payload of the timed
    pass is compared (SHA-256) with the reference's output for that image (tests/golden/cfg5.json, no oracle in the loop).
    Returns (seconds, texels, info)."""
    import directxtex_amd as dx
    from directxtex_amd import sharding, synth
    side = CFG5_SIDE
    distinct = {}

    def load(i):
        if i % 16 not in distinct:
            distinct[i % 16] = synth.survey_rgba8(side, side, 1000 + (i % 16), "opaque")
        return distinct[i % 16]

    mine = sharding.images_for_rank(1024, world, rank)[:per_rank]
    for i in mine:
        load(i)                                           # image synthesis is not part of the measurement
    many = lambda imgs: ctx.compress_many(imgs, side, side, dx.DXGI_FORMAT_R8G8B8A8_UNORM, dx.DXGI_FORMAT_BC7_UNORM, 0, 0.5)
    sharding.run_shard(1024, world, rank, load, many, batch=128, limit=min(per_rank, 16))       # untimed, two chunks: allocates both lanes' pinned / device staging and the copy streams
    t0 = time.perf_counter()
    res = sharding.run_shard(1024, world, rank, load, many, batch=128, limit=per_rank)
    dt = time.perf_counter() - t0
    assert sorted(res) == mine
    per_chunk = max(1, CFG5_CHUNK_TEXELS // (side * side))
    info = {"indices": mine, "chunks": (len(mine) + per_chunk - 1) // per_chunk}
    gp = os.path.join(ROOT, "tests", "golden", "cfg5.json")
    if os.path.exists(gp):
        gold = json.load(open(gp))["images"]
        checked = same = 0
        for i, payload in res.items():
            g = gold.get(str(i % 16))
            if g and hashlib.sha256(distinct[i % 16].tobytes()).hexdigest() == g["input_sha256"]:
                checked += 1
                same += hashlib.sha256(payload.tobytes()).hexdigest() == g["sha256"]
        info["payloads_checked"] = checked
        info["payloads_identical"] = same
        info["ref_seconds_per_image_golden"] = round(float(np.mean([v["ref_seconds"] for v in gold.values()])), 1)
    return dt, float(len(mine)) * side * side, info


def cfg5_cpu_side(budget_s=8.0):
    """SURVEY 8d's CPU side of cfg5: the reference on cfg5 image 0 (seed 1000), here on bands of 16 block rows for ~budget_s, the
    rate extrapolated to the 1024 images and labelled so."""
    import oracle
    from directxtex_amd import synth
    if not oracle.have_ref():
        return None
    img = synth.survey_rgba8(CFG5_SIDE, CFG5_SIDE, 1000, "opaque")
    rows = BAND_ROWS * 4
    nb = CFG5_SIDE // rows
    secs = 0.0; texels = 0; k = 0
    while k < nb and (k == 0 or secs + secs / k < budget_s):
        b = (k * 7 + nb // 2) % nb                         # bands spread over the image
        crop = np.ascontiguousarray(img[b * rows:(b + 1) * rows])
        t0 = time.perf_counter()
        oracle.ref_compress_image(crop, CFG5_SIDE, rows, 28, 98, TEX_COMPRESS_PARALLEL, 0.5)
        secs += time.perf_counter() - t0
        texels += rows * CFG5_SIDE; k += 1
    rate = texels / secs / 1e6
    return {"value": round(rate, 5), "unit": "Mtexels/s", "cores": oracle.ref_num_threads(), "kind": "reference",
            "sample": f"{k} bands of {rows} rows of cfg5 image 0 ({texels} texels, {secs:.1f} s)",
            "extrapolated_seconds_for_1024_images": round(1024.0 * CFG5_SIDE * CFG5_SIDE / (rate * 1e6), 0)}


def end_to_end(ctx, img, gold_sha):
    """The headline image through dxtex_compress (host pointers: H2D + kernels + D2H, what texconv times around its Compress call,
    Texconv/texconv.cpp:3692-3712): from pageable numpy memory and from pinned memory, best of two each."""
    import torch
    import directxtex_amd as dx
    out = {}
    rp, sp = dx.compute_pitch(dx.DXGI_FORMAT_BC7_UNORM, WIDTH, HEIGHT)
    pin_src = torch.empty(img.nbytes, dtype=torch.uint8).pin_memory()
    pin_src.numpy()[:] = img.reshape(-1).view(np.uint8)
    pin_dst = torch.empty(sp, dtype=torch.uint8).pin_memory()
    for name, src_arr, dst_arr in (("pageable", img, np.zeros(sp, np.uint8)), ("pinned", pin_src.numpy(), pin_dst.numpy())):
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            ctx.compress_into(src_arr, WIDTH, HEIGHT, dx.DXGI_FORMAT_R8G8B8A8_UNORM, dst_arr, dx.DXGI_FORMAT_BC7_UNORM, 0, 0.5)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        e = {"ms": round(best * 1e3, 3), "Mtexels_s": round(WIDTH * HEIGHT / best / 1e6, 2)}
        if gold_sha:
            e["identical_to_reference_golden"] = hashlib.sha256(dst_arr.tobytes()).hexdigest() == gold_sha
        out[name] = e
    out["note"] = "dxtex_compress: 64 MiB host -> device, kernels, 16 MiB device -> host, one image, synchronous call"
    return out


# kernel name the context's profiler reports -> name rocprofv3 reports
ROCPROF_NAME = {"bc7_exhaustive_mode1": "bc7_exhaustive_kernel<1, 0, 0>", "bc7_exhaustive_mode3": "bc7_exhaustive_kernel<3, 0, 0>",
                "bc7_perturb_mode1": "bc7_perturb_filter_kernel<1, 0, 0>", "bc7_perturb_mode3": "bc7_perturb_kernel<3, 0, 0>", "bc7_rough": "bc7_rough_kernel"}


def live_pmc(dom, budget_s=150.0):
    """PMC counters of the dominant kernel, collected NOW, in this run: bench.py cannot read counters in its own process, so it runs the
    same workload (tools/prof_workloads.py bc7: the same image, the same call, one repetition) under `rocprofv3 --kernel-trace --pmc ...`
    in child processes - separate passes for TCC FETCH_SIZE, TCC WRITE_SIZE (they do not fit one pass) and the SQ VALU counters, as
    MI355X_MICROARCH.md's rocprofv3 section prescribes - and parses the CSVs. Returns None when rocprofv3 is missing or a pass fails
    (the committed profiles/pmc_traffic.json is used then, and says so)."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    name = ROCPROF_NAME.get(dom)
    if not name or not os.path.exists(exe):
        return None
    t_start = time.perf_counter()
    out = {}
    tmp = tempfile.mkdtemp(prefix="dxtex_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    passes = (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("sq", ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]),
              ("lds", ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]))
    try:
        for tag, counters in passes:
            if time.perf_counter() - t_start > budget_s:
                return None
            cmd = [exe, "--kernel-trace", "--pmc"] + counters + ["-d", tmp, "-o", tag, "--output-format", "csv", "--",
                                                                  sys.executable, os.path.join(ROOT, "tools", "prof_workloads.py"), "bc7", "1"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=180)
            files = glob.glob(os.path.join(tmp, "**", tag + "_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            acc = {}; n = {}; dur = []
            for row in csv.DictReader(open(files[0])):
                if tag in ("fetch", "write") and "bc7_" in row["Kernel_Name"]:      # every launch of the image (one repetition): the whole step's traffic
                    out["STEP_" + row["Counter_Name"]] = out.get("STEP_" + row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                if name in row["Kernel_Name"]:
                    c = row["Counter_Name"]
                    acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"]); n[c] = n.get(c, 0) + 1
                    dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
            if not acc:
                if tag == "lds":
                    continue                                  # optional pass
                return None
            for c in acc:
                out[c] = acc[c] / n[c]
            out.setdefault("ms_under_counters", {})[tag] = round(sum(dur) / len(dur), 3)
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out["seconds"] = round(time.perf_counter() - t_start, 1)
    return out


_GOLD = None


def golden_case(cid):
    global _GOLD
    if _GOLD is None:
        p = os.path.join(ROOT, "tests", "golden", "fullsize.json")
        _GOLD = json.load(open(p)) if os.path.exists(p) else {"cases": {}}
    return _GOLD["cases"].get(cid)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full", action="store_true", help="run the reference on the whole 4096^2 image (about 3.5 min on 128 threads)")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 --pmc child passes for roofline.traffic / roofline.valu (the committed "
                    "profiles/pmc_traffic.json is used instead, if it matches the kernel sources)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads (cfg3 / cfg4 / cfg5 shard / BC1-5 / decode) reported next to the headline")
    ap.add_argument("--cfg5-images", type=int, default=128, help="images of the cfg5 shard every rank compresses after the timed region (0 = skip); "
                    "128 = a rank's whole share of the 1024 images at 8 GPUs, sixteen chunks of dxtex_compress_many's double-buffered pipeline")
    ap.add_argument("--inprocess-contexts", type=int, default=0, help="contexts of the in-process split leg (dxtex_compress_multi); 0 = one per visible GPU, "
                    "reported when there are at least two")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo lets several "
                    "ranks share one GPU in the tests)")
    args = ap.parse_args()

    import torch
    import directxtex_amd as dx

    from directxtex_amd import sharding
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())     # more ranks than GPUs (gloo tests): ranks share a GPU
    torch.cuda.set_device(local_rank)
    rank, world = sharding.init_from_env(args.backend, torch.device("cuda", local_rank))      # "nccl" is RCCL on ROCm
    distributed = world > 1
    n_gpus = world if distributed else 1
    if args.gpus != n_gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={n_gpus}", file=sys.stderr)

    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    ctx = dx.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)

    img = make_image(rank)                                # each GPU compresses its own image
    src = torch.from_numpy(img).to(dev)
    rp, sp = dx.compute_pitch(dx.DXGI_FORMAT_BC7_UNORM, WIDTH, HEIGHT)
    dst = torch.empty(sp, dtype=torch.uint8, device=dev)

    def step():
        ctx.compress_device(src.data_ptr(), WIDTH, HEIGHT, dx.DXGI_FORMAT_R8G8B8A8_UNORM,
                            dst.data_ptr(), dx.DXGI_FORMAT_BC7_UNORM, dx.TEX_COMPRESS_DEFAULT, 0.5)

    def barrier():
        sharding.barrier(world)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)

    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0

    # whole-job throughput: texels of all ranks / slowest rank's time (no data-path collective anywhere)
    elapsed, texels = sharding.aggregate(elapsed, float(WIDTH) * HEIGHT * args.steps, world, dev)
    value = texels / elapsed / 1e6

    # per-kernel durations: a separate, untimed pass with hipEvents around every launch on the launch stream
    ctx.profile_begin()
    nprof = 2
    for _ in range(nprof):
        step()
    kernels = ctx.profile_end()

    # cfg5 shard: every rank, after the timed region (reported next to the headline; its own max-over-ranks timing)
    cfg5 = None
    if args.cfg5_images > 0 and not args.no_extra:
        barrier()
        try:
            dt5, tex5, info5 = cfg5_shard(ctx, dev, rank, world, args.cfg5_images)
            dt5, tex5 = sharding.aggregate(dt5, tex5, world, dev)
            info5 = sharding.gather_objects(info5, world)          # bookkeeping only: which indices ran where, digests checked
            checked = sum(i.get("payloads_checked", 0) for i in info5); same = sum(i.get("payloads_identical", 0) for i in info5)
            all_idx = sorted(j for i in info5 for j in i["indices"])
            cfg5 = {"images": int(round(tex5 / (2048 * 2048))), "seconds": round(dt5, 3), "Mtexels_s": round(tex5 / dt5 / 1e6, 2),
                    "chunks": min(i["chunks"] for i in info5), "indices_disjoint": len(set(all_idx)) == len(all_idx),
                    "indices_per_rank": [i["indices"][:4] + (["..."] if len(i["indices"]) > 4 else []) for i in info5],
                    "payloads_checked": checked, "payloads_identical": same,
                    "identical_to_reference_golden": bool(checked) and checked == same and checked == int(round(tex5 / (2048 * 2048))),
                    "golden": "tests/golden/cfg5.json: SHA-256 of the reference's payload for each of the 16 distinct images",
                    "workload": f"cfg5 shard: images i = rank (mod {world}) of 1024 x 2048^2 RGBA8 -> BC7, the first {args.cfg5_images} per GPU, host buffers in and "
                                "out through dxtex_compress_many (PCIe-inclusive), chunks of eight images",
                    "roofline": hbm_roofline(tex5 * 5.0, dt5 * 1e3, time_key="wall_ms")}
            if info5[0].get("ref_seconds_per_image_golden"):
                cfg5["reference_seconds_per_image_8_threads"] = info5[0]["ref_seconds_per_image_golden"]
        except Exception as e:
            cfg5 = {"error": repr(e)}
            if distributed:
                raise

    # One image split over the GPUs (strong scaling): every rank encodes a stripe of block rows of THE SAME 4096^2 image (rank 0's),
    # the stripes are gathered on every rank - the package's only data-path collective, one all_gather of 16 MiB / N per rank. Reported next
    # to the headline for N > 1; the headline itself stays image-per-GPU.
    split = None
    if distributed and not args.no_extra:
        try:
            img0 = make_image(0)
            nbh = HEIGHT // 4
            r0, r1 = sharding.stripe_rows(nbh, world, rank)
            s_src = torch.from_numpy(np.ascontiguousarray(img0[r0 * 4:r1 * 4])).to(dev)
            s_dst = torch.empty((r1 - r0) * rp, dtype=torch.uint8, device=dev)

            def split_step():
                if r1 > r0:
                    ctx.compress_device(s_src.data_ptr(), WIDTH, (r1 - r0) * 4, dx.DXGI_FORMAT_R8G8B8A8_UNORM, s_dst.data_ptr(), dx.DXGI_FORMAT_BC7_UNORM, 0, 0.5)
                torch.cuda.synchronize(dev)
                return sharding.gather_stripes(s_dst, nbh, rp, world, rank)

            split_step()                                      # warm-up: scratch for the stripe's size, the collective's buffers
            barrier(); torch.cuda.synchronize(dev)
            ts = time.perf_counter()
            nsplit = max(1, min(3, args.steps))
            for _ in range(nsplit):
                whole = split_step()
            torch.cuda.synchronize(dev); barrier()
            dts, _ = sharding.aggregate((time.perf_counter() - ts) / nsplit, 0.0, world, dev)
            if rank == 0:
                gold = golden_case("cfg2_bc7_4096")
                ok = None
                if gold and hashlib.sha256(img0.tobytes()).hexdigest() == gold["input_sha256"]:
                    ok = band_sha(whole.cpu().numpy(), WIDTH, HEIGHT) == gold["bands"]
                split = {"ms": round(dts * 1e3, 3), "Mtexels_s": round(WIDTH * HEIGHT / dts / 1e6, 2), "n_gpus": world, "scaling": "strong",
                         "identical_to_reference_golden": ok,
                         "workload": f"ONE 4096^2 RGBA8 image -> BC7: rank r encodes block rows [r, r+1) * {nbh} / {world}, the stripes are all_gathered ({args.backend}); "
                                     "time = slowest rank, payload on every GPU when it stops"}
            del s_src, s_dst, whole
        except Exception as e:
            if rank == 0:
                split = {"error": repr(e)}
            raise            # a rank that stopped here would leave the others waiting in the section's collectives until the backend times out

    # The same split inside ONE process (dxtex_compress_multi: a context per visible GPU, a thread per context, stripes of block rows of a host
    # image): what a caller of the host layer's Compress(devices, n, ...) gets. Rank 0 drives every GPU while the other ranks wait, so it is
    # reported next to the rank-per-GPU split, not instead of it. Host memory in and out: PCIe-inclusive.
    inproc = None
    ngpu = torch.cuda.device_count()
    ndev = args.inprocess_contexts if args.inprocess_contexts > 0 else ngpu       # (--inprocess-contexts N: N contexts dealt over the visible GPUs, for boxes with one GPU)
    if ndev > 1 and not args.no_extra:
        barrier()
        if rank == 0:
            try:
                img0 = make_image(0)
                extra_ctxs = [dx.Context((local_rank + i) % ngpu) for i in range(1, ndev)]
                all_ctxs = [ctx] + extra_ctxs
                out = np.zeros(sp, np.uint8)
                times = {}
                for n in sorted({1, 2, ndev}):
                    dx.capi.compress_multi(all_ctxs[:n], img0, WIDTH, HEIGHT, dx.DXGI_FORMAT_R8G8B8A8_UNORM, dx.DXGI_FORMAT_BC7_UNORM, 0, 0.5, out=out)     # warm-up: scratch and staging of the stripe's size
                    best = 1e9
                    for _ in range(2):
                        t1 = time.perf_counter()
                        dx.capi.compress_multi(all_ctxs[:n], img0, WIDTH, HEIGHT, dx.DXGI_FORMAT_R8G8B8A8_UNORM, dx.DXGI_FORMAT_BC7_UNORM, 0, 0.5, out=out)
                        best = min(best, time.perf_counter() - t1)
                    times[n] = best
                gold = golden_case("cfg2_bc7_4096")
                ok = None
                if gold and hashlib.sha256(img0.tobytes()).hexdigest() == gold["input_sha256"]:
                    ok = band_sha(out, WIDTH, HEIGHT) == gold["bands"]
                inproc = {"ms": round(times[ndev] * 1e3, 3), "Mtexels_s": round(WIDTH * HEIGHT / times[ndev] / 1e6, 2), "n_contexts": ndev, "n_gpus_visible": ngpu, "scaling": "strong",
                          "ms_by_contexts": {str(k): round(v * 1e3, 3) for k, v in times.items()}, "identical_to_reference_golden": ok,
                          "workload": "ONE 4096^2 RGBA8 image in host memory -> BC7 through dxtex_compress_multi: a context per GPU in this process, a thread per context, "
                                      "stripes of block rows; upload, encode and download inside the time"}
                # the mip chain of the same image split the same way (dxtex_generate_mips_multi: resident stripes - one upload and one download per
                # context and chain, nothing between the contexts), with what every context moved over the host link
                try:
                    chain = {}
                    for n in sorted({1, ndev}):
                        use = all_ctxs[:n]
                        dx.capi.generate_mips_multi(use, img0, WIDTH, HEIGHT, dx.DXGI_FORMAT_R8G8B8A8_UNORM, 13, 0x300000)
                        for c in use:
                            c.transfer_bytes(reset=True)
                        t1 = time.perf_counter()
                        lv = dx.capi.generate_mips_multi(use, img0, WIDTH, HEIGHT, dx.DXGI_FORMAT_R8G8B8A8_UNORM, 13, 0x300000)
                        dt = time.perf_counter() - t1
                        moved = [c.transfer_bytes() for c in use]
                        chain[str(n)] = {"ms": round(dt * 1e3, 3), "h2d_bytes_per_context": [m[0] for m in moved], "d2h_bytes_per_context": [m[1] for m in moved],
                                         "levels_sha256_16": hashlib.sha256(b"".join(x.tobytes() for x in lv)).hexdigest()[:16]}
                    inproc["mip_chain_cubic_4096"] = {"by_contexts": chain, "identical_across_context_counts": len({v["levels_sha256_16"] for v in chain.values()}) == 1,
                                                      "source_bytes": WIDTH * HEIGHT * 4,
                                                      "workload": "4096^2 RGBA8 host image -> 13-level cubic chain in host memory through dxtex_generate_mips_multi"}
                except Exception as e:
                    inproc["mip_chain_cubic_4096"] = {"error": repr(e)}
                for c in extra_ctxs:
                    c.close()
            except Exception as e:
                inproc = {"error": repr(e)}
        barrier()

    if rank == 0:
        # ---- roofline of the dominant kernel -----------------------------------------------------------
        per_launch = {k: (ms / max(1, n)) for k, (ms, n) in kernels.items()}
        dom = max(per_launch, key=per_launch.get) if per_launch else None
        algo_bytes = ALGO_BYTES_PER_TEXEL * WIDTH * HEIGHT
        roof = None
        if dom:
            roof = hbm_roofline(algo_bytes, per_launch[dom], dom)
            roof["note"] = ("BC7 at the reference's search depth is VALU-bound (integer endpoint search); the HBM fraction is reported as the "
                            "contract asks, the VALU issue utilisation of the same kernel is under `valu`")
            roof["all_kernels_ms"] = {k: round(v, 4) for k, v in sorted(per_launch.items(), key=lambda kv: -kv[1])}
            roof["step_kernel_ms"] = round(sum(ms for ms, n in kernels.values()) / nprof, 4)
            roof["step_hbm_frac"] = round(algo_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 6)
            # the late phase (modes 4 / 5 / 6 on the blocks the earlier modes left): its kernels one after the other (the profiling pass is serial)
            roof["late_phase_kernel_ms"] = round(sum(v for k, v in per_launch.items() if k.endswith("_late")), 3)
        # PMC counters: measured in THIS run by child rocprofv3 passes over the same workload (live_pmc); the committed file only supplies the
        # static issue cost per instruction of the kernel's opcode mix (a property of the code: it needs a disassembly) and is the fallback
        # for everything when rocprofv3 is not available - and then ONLY if it was measured on these kernel sources (stamp) and names this
        # run's dominant kernel; otherwise traffic stays null.
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        committed = None
        if os.path.exists(pmc):
            try:
                committed = json.load(open(pmc))
            except Exception:
                committed = None
        live = None
        if roof and n_gpus == 1 and not args.no_pmc and not args.no_extra:
            live = live_pmc(dom)
        if roof and live and "FETCH_SIZE" in live and "WRITE_SIZE" in live:
            # rocprofv3 reports the TCC sizes in KiB; the search kernels read task records and 4-byte texels, not 16-byte-per-lane image streams,
            # so FETCH_SIZE needs no x2 correction here (MI355X_MICROARCH.md, HBM section)
            roof["traffic"] = int((live["FETCH_SIZE"] + live["WRITE_SIZE"]) * 1024)
            roof["traffic_source"] = "this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate child passes over tools/prof_workloads.py bc7 (same image, same call)"
            roof["traffic_over_algorithmic"] = round(roof["traffic"] / algo_bytes, 2)
            if live.get("SQ_INSTS_VALU"):
                same_src = bool(committed) and committed.get("sources_sha256") == kernel_sources_sha256() and committed.get("kernel") == dom
                cyc = committed["valu"]["mean_issue_cycles_per_inst"] if same_src and committed.get("valu") else None
                v = {"valu_insts_per_launch": int(live["SQ_INSTS_VALU"]), "waves": int(live.get("SQ_WAVES", 0)),
                     "active_lanes_per_valu_inst": round(live["SQ_THREAD_CYCLES_VALU"] / max(1.0, live["SQ_ACTIVE_INST_VALU"]), 1),
                     "kernel_ms_live": round(per_launch[dom], 3), "kernel_ms_under_counters": live.get("ms_under_counters"),
                     "method": "SQ_INSTS_VALU (this run) per SIMD x loop-weighted static issue cost of the kernel's opcode mix (profiles/r02_valu_rates.md) / (live duration x 2.4 GHz)"}
                if cyc:
                    v["mean_issue_cycles_per_inst"] = cyc
                    v["issue_utilisation"] = round(min(1.0, live["SQ_INSTS_VALU"] / 1024.0 * cyc / (per_launch[dom] * 1e-3 * 2.4e9)), 3)
                else:
                    v["issue_utilisation_if_2_or_4_cycles"] = [round(live["SQ_INSTS_VALU"] / 1024.0 * c / (per_launch[dom] * 1e-3 * 2.4e9), 3) for c in (2.15, 4.15)]
                tot = live.get("SQ_WAIT_ANY", 0) + live.get("SQ_WAIT_INST_ANY", 0) + live.get("SQ_ACTIVE_INST_ANY", 0)
                if tot:
                    v["wave_cycles_parked_stalled_issuing"] = [round(live.get(k, 0) / tot, 2) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")]
                roof["valu"] = v
                # the same evidence as flat scalars (nested objects do not survive every consumer of this line)
                if "issue_utilisation" in v:
                    roof["valu_issue_utilisation"] = v["issue_utilisation"]
                roof["active_lanes_per_valu_inst"] = v["active_lanes_per_valu_inst"]
            if live.get("SQ_LDS_IDX_ACTIVE"):
                roof["lds_bank_conflict_share"] = round(live.get("SQ_LDS_BANK_CONFLICT", 0.0) / live["SQ_LDS_IDX_ACTIVE"], 3)
            if "STEP_FETCH_SIZE" in live and "STEP_WRITE_SIZE" in live:
                roof["traffic_whole_step"] = int((live["STEP_FETCH_SIZE"] + live["STEP_WRITE_SIZE"]) * 1024)
                roof["traffic_whole_step_over_algorithmic"] = round(roof["traffic_whole_step"] / algo_bytes, 1)
            roof["pmc_seconds"] = live.get("seconds")
        elif roof and os.path.exists(pmc):
            try:
                t = json.load(open(pmc))
                if t.get("kernel") != dom:
                    roof["traffic_note"] = f"profiles/pmc_traffic.json describes {t.get('kernel')}, not this run's dominant kernel: dropped"
                elif t.get("sources_sha256") != kernel_sources_sha256():
                    roof["traffic_note"] = "profiles/pmc_traffic.json was measured on other kernel sources (sources_sha256 differs): dropped"
                else:
                    roof["traffic"] = t.get("hbm_bytes_per_launch")
                    roof["traffic_source"] = t.get("source")
                    if t.get("valu"):
                        roof["valu"] = t["valu"]               # SQ counters of the same kernel (profiles/): what actually bounds it
                        for k_ in ("issue_utilisation", "active_lanes_per_valu_inst", "lds_bank_conflict_share"):
                            if k_ in t["valu"]:
                                roof["valu_issue_utilisation" if k_ == "issue_utilisation" else k_] = t["valu"][k_]
                    if t.get("traffic_whole_step_over_algorithmic"):
                        roof["traffic_whole_step_over_algorithmic"] = t["traffic_whole_step_over_algorithmic"]
            except Exception:
                pass

        # ---- parity + quality + CPU baseline (N = 1 only) --------------------------------------------------
        cpu = None
        parity = {}
        extra = {}
        if n_gpus == 1:
            try:
                payload = dst.cpu().numpy()
                gold = golden_case("cfg2_bc7_4096")
                if gold and hashlib.sha256(img.tobytes()).hexdigest() == gold["input_sha256"]:
                    bands = band_sha(payload, WIDTH, HEIGHT)
                    parity["golden"] = "tests/golden/fullsize.json: reference output for the whole 4096^2 image (1 048 576 blocks)"
                    parity["golden_bands_identical"] = sum(1 for a, b in zip(bands, gold["bands"]) if a == b)
                    parity["golden_bands"] = len(gold["bands"])
                    parity["full_image_identical_to_reference"] = bands == gold["bands"]
                    parity["reference_psnr_db_full_image"] = gold["psnr_db"]
                parity["gpu_psnr_db_full_image"] = round(gpu_psnr(ctx, dev, src, dst, 28, 98, WIDTH, HEIGHT), 4)
                if not args.no_cpu_baseline:
                    cpu, live = cpu_baseline(img, payload, args.cpu_full)
                    parity.update(live)
                    # the whole image on the reference: measured by the live parity test on the GPU box (tests/test_zz_fullsize_gpu.py),
                    # committed under profiles/ (3+ minutes - not re-run here; --cpu-full does)
                    for name in ("r06_fullsize_live.json", "r05_fullsize_live.json", "r03_fullsize_live.json", "r02_fullsize_live.json"):
                        fp = os.path.join(ROOT, "profiles", name)
                        if cpu and os.path.exists(fp):
                            w = json.load(open(fp)).get("cfg2_bc7_4096")
                            if w:
                                cpu["whole_image"] = {"seconds": w["ref_seconds"], "cores": w["ref_threads"], "Mtexels_s": w["Mtexels_s"],
                                                      "identical_to_gpu": w["identical"], "source": f"profiles/{name} (live parity run, same image)"}
                                # flat scalars (nested objects do not survive every parser of this line)
                                extra["cpu_baseline_whole_image_Mtexels_s"] = w["Mtexels_s"]
                                extra["cpu_baseline_sample_over_whole_image"] = round(cpu["value"] / w["Mtexels_s"], 4)
                                break
                if not args.no_extra:
                    extra["end_to_end"] = end_to_end(ctx, img, gold["sha256"] if gold else None)
            except Exception as e:                              # the baseline must never break the bench line
                extra["cpu_baseline_error"] = repr(e)

        if n_gpus == 1 and not args.no_extra:
            try:
                extra["other_workloads"] = other_workloads(ctx, dev, img, rank, world)
            except Exception as e:
                extra["other_workloads_error"] = repr(e)
        if split:
            extra.setdefault("other_workloads", {})["bc7_4096_split"] = split
        if inproc:
            extra.setdefault("other_workloads", {})["bc7_4096_inprocess_split"] = inproc
        if cfg5:
            if n_gpus == 1 and not args.no_cpu_baseline and "error" not in cfg5:
                try:
                    cfg5["cpu_baseline"] = cfg5_cpu_side()
                except Exception as e:
                    cfg5["cpu_baseline_error"] = repr(e)
            extra.setdefault("other_workloads", {})["cfg5_shard"] = cfg5

        line = {
            "metric": "Mtexels/s BC7 encode (4096^2 RGBA8, TEX_COMPRESS_DEFAULT)",
            "value": round(value, 3), "unit": "Mtexels/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp32 seed fit + i32 exact error sums (u8 texels in, 128-bit blocks out)", "data": "synthetic",
            "config": {"workload": "cfg2: 4096x4096 RGBA8 -> BC7_UNORM, TEX_COMPRESS_DEFAULT, one image per GPU, source and "
                                   "payload resident in HBM (dxtex_compress_device)",
                       "image": "SURVEY.md 8d cfg2 recipe (directxtex_amd.synth.survey_rgba8: LCG gradients + 4-octave noise, flat to noisy blocks), "
                                "opaque, seed 2+rank",
                       "sharding": f"image-per-GPU x{n_gpus}, no data-path collective"},
            "roofline": roof, "cpu_baseline": cpu, "parity": parity,
        }
        line.update(extra)
        print(json.dumps(line), flush=True)

    ctx.close()
    if distributed:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
