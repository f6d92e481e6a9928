#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X DirectXTex hot path.

Metric (BASELINE.json): Mtexels/s of BC7 encode, 4096x4096 RGBA8, TEX_COMPRESS_DEFAULT.

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of DirectX::Compress' hot path (dxtex_compress_device: source resident in HBM, BC7
payload written to HBM) over one 4096^2 synthetic image per GPU. N > 1 is launched by the driver through
torch.distributed.run, one rank per GPU; images are sharded one-per-GPU (the path has no exchange step, so
there is no data-path collective: the only collectives are the timing barrier and the MAX over ranks).
Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     - dominant kernel: algorithmic bytes per launch / its mean duration (hipEvents on the launch
                 stream, recorded inside the timed region through dxtex_ctx_profile_*), against 8 TB/s HBM.
                 BC7 at the reference's search depth is VALU-bound, so the HBM fraction is small by design;
                 `all_kernels` lists every kernel of the step.
  cpu_baseline - the reference's own encoder (oracle/_ref, D3DXEncodeBC7 compiled in place, OpenMP over
                 blocks as CompressBC_Parallel does) timed on this box's host cores on a bounded sample of
                 the same image (rank 0, N = 1 only). Reported, not a target.
"""
import argparse
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


def make_image(seed):
    """4096^2 RGBA8 synthetic texture (opaque), SURVEY.md section 8d recipe. Built from a 1024^2 hash-noise
    image generated at 4 different seeds and tiled 4x4, so generation stays a few seconds."""
    from directxtex_amd import synth
    tiles = [synth.rgba8(1024, 1024, seed=seed * 16 + i, alpha="opaque") for i in range(4)]
    rows = []
    for y in range(4):
        rows.append(np.concatenate([tiles[(x + y) % 4] for x in range(4)], axis=1))
    return np.ascontiguousarray(np.concatenate(rows, axis=0))


def cpu_baseline(img, budget_s=15.0):
    """Reference encoder on the host cores over a bounded crop of the benchmark image."""
    import oracle
    if not oracle.have_ref():
        return None
    fmt_src, fmt_bc7 = 28, 98
    threads = oracle.ref_num_threads()
    # calibrate on 8x8 blocks, then size the sample for ~budget_s of wall time
    crop = np.ascontiguousarray(img[:32, :32])
    t0 = time.perf_counter()
    oracle.compress_image(crop, 32, 32, fmt_src, fmt_bc7, 0, 0.5)
    dt = max(time.perf_counter() - t0, 1e-4)
    blocks = int(max(64, min(65536, 64 * budget_s / dt)))
    side = int(np.sqrt(blocks)) * 4
    side = max(32, min(1024, side // 32 * 32))
    y0 = x0 = 1024 - side // 2                       # a crop that straddles flat, noisy and edge regions
    crop = np.ascontiguousarray(img[y0:y0 + side, x0:x0 + side])
    t0 = time.perf_counter()
    payload = oracle.compress_image(crop, side, side, fmt_src, fmt_bc7, 0, 0.5)
    dt = time.perf_counter() - t0
    src = oracle.load_image(crop, side, side, fmt_src)
    psnr = oracle.psnr_rgb(oracle.decode_image(payload, side, side, fmt_bc7)[..., :3], src[..., :3])
    return {"value": round(side * side / dt / 1e6, 5), "unit": "Mtexels/s", "cores": threads, "kind": "reference",
            "sample": f"{side}x{side} crop at ({x0},{y0}) of the benchmark image, D3DXEncodeBC7 flags=0, "
                      f"OpenMP over blocks, {dt:.1f} s", "psnr_db": round(psnr, 3)}, (x0, y0, side, payload)


def other_workloads(ctx, dev, img):
    """The other configurations of BASELINE.json, measured once each AFTER the timed region (reported, not the metric):
    device-resident inputs, per-call wall time with a stream sync, algorithmic GB/s per SURVEY.md section 8d."""
    import torch
    import directxtex_amd as dx
    RGBA8, RGBA16F = dx.DXGI_FORMAT_R8G8B8A8_UNORM, dx.DXGI_FORMAT_R16G16B16A16_FLOAT
    out = {}

    def timed(fn, n):
        fn(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n

    src = torch.from_numpy(img).to(dev)
    for name, fmt, bpt, n in (("bc1", dx.DXGI_FORMAT_BC1_UNORM, 4.5, 20), ("bc3", dx.DXGI_FORMAT_BC3_UNORM, 5.0, 20), ("bc5", dx.DXGI_FORMAT_BC5_UNORM, 5.0, 20)):
        rp, sp = dx.compute_pitch(fmt, WIDTH, HEIGHT)
        dst = torch.empty(sp, dtype=torch.uint8, device=dev)
        dt = timed(lambda: ctx.compress_device(src.data_ptr(), WIDTH, HEIGHT, RGBA8, dst.data_ptr(), fmt, 0, 0.5), n)
        out[f"{name}_4096"] = {"ms": round(dt * 1e3, 3), "Mtexels_s": round(WIDTH * HEIGHT / dt / 1e6, 1), "algorithmic_GBs": round(WIDTH * HEIGHT * bpt / dt / 1e9, 1)}
    # the reference's faster / slower BC7 settings on the same image (TEX_COMPRESS_BC7_QUICK: mode 6 only; BC7_USE_3SUBSETS: + modes 0, 2)
    rp, sp = dx.compute_pitch(dx.DXGI_FORMAT_BC7_UNORM, WIDTH, HEIGHT)
    dst = torch.empty(sp, dtype=torch.uint8, device=dev)
    for name, fl, n in (("bc7_quick_4096", dx.TEX_COMPRESS_BC7_QUICK, 5), ("bc7_3subsets_4096", 0x80000, 1)):
        dt = timed(lambda: ctx.compress_device(src.data_ptr(), WIDTH, HEIGHT, RGBA8, dst.data_ptr(), dx.DXGI_FORMAT_BC7_UNORM, fl, 0.5), n)
        out[name] = {"ms": round(dt * 1e3, 2), "Mtexels_s": round(WIDTH * HEIGHT / dt / 1e6, 1)}
    # cfg3: 4096^2 RGBA16F -> BC6H_UF16
    hdr = torch.from_numpy((img.astype(np.float32) * (8.0 / 255.0)).astype(np.float16)).to(dev)
    rp, sp = dx.compute_pitch(dx.DXGI_FORMAT_BC6H_UF16, WIDTH, HEIGHT)
    dst = torch.empty(sp, dtype=torch.uint8, device=dev)
    dt = timed(lambda: ctx.compress_device(hdr.data_ptr(), WIDTH, HEIGHT, RGBA16F, dst.data_ptr(), dx.DXGI_FORMAT_BC6H_UF16, 0, 0.5), 2)
    out["bc6h_uf16_4096"] = {"ms": round(dt * 1e3, 2), "Mtexels_s": round(WIDTH * HEIGHT / dt / 1e6, 2), "algorithmic_GBs": round(WIDTH * HEIGHT * 9.0 / dt / 1e9, 2)}
    # decode BC7 4096^2 -> RGBA8 (0.5 + ... 1 B read + 4 B written per texel)
    rp7, sp7 = dx.compute_pitch(dx.DXGI_FORMAT_BC7_UNORM, WIDTH, HEIGHT)
    bc7 = torch.empty(sp7, dtype=torch.uint8, device=dev)
    ctx.compress_device(src.data_ptr(), WIDTH, HEIGHT, RGBA8, bc7.data_ptr(), dx.DXGI_FORMAT_BC7_UNORM, dx.TEX_COMPRESS_BC7_QUICK, 0.5)
    back = torch.empty(WIDTH * HEIGHT * 4, dtype=torch.uint8, device=dev)
    dt = timed(lambda: ctx.decompress_device(bc7.data_ptr(), WIDTH, HEIGHT, dx.DXGI_FORMAT_BC7_UNORM, back.data_ptr(), RGBA8), 20)
    out["bc7_decode_4096"] = {"ms": round(dt * 1e3, 3), "Mtexels_s": round(WIDTH * HEIGHT / dt / 1e6, 1), "algorithmic_GBs": round(WIDTH * HEIGHT * 5.0 / dt / 1e9, 1)}
    # cfg4: 8192^2 RGBA8 full mip chain (box, cubic) then BC3 of all 14 levels
    big = src.reshape(HEIGHT, WIDTH, 4).repeat(2, 2, 1).contiguous()
    w = h = 8192
    sizes = []
    while True:
        sizes.append((w, h))
        if w == 1 and h == 1:
            break
        w, h = max(1, w >> 1), max(1, h >> 1)
    bufs = [big.reshape(-1)] + [torch.empty(a * b * 4, dtype=torch.uint8, device=dev) for a, b in sizes[1:]]
    levels = [dx.capi.device_image(t.data_ptr(), a, b, RGBA8) for t, (a, b) in zip(bufs, sizes)]
    chain_bytes = sum(a * b * 4 for a, b in sizes[:-1]) + sum(a * b * 4 for a, b in sizes[1:])
    for name, flt in (("box", dx.TEX_FILTER_BOX), ("cubic", dx.TEX_FILTER_CUBIC)):
        dt = timed(lambda: ctx.generate_mips_device(levels, flt), 5)
        out[f"mips_{name}_8192"] = {"ms": round(dt * 1e3, 3), "algorithmic_GBs": round(chain_bytes / dt / 1e9, 1)}
    bc3 = [torch.empty(dx.compute_pitch(dx.DXGI_FORMAT_BC3_UNORM, a, b)[1], dtype=torch.uint8, device=dev) for a, b in sizes]
    dsts = [dx.capi.device_image(t.data_ptr(), a, b, dx.DXGI_FORMAT_BC3_UNORM) for t, (a, b) in zip(bc3, sizes)]
    dt = timed(lambda: ctx.compress_many_device(levels, dsts, 0, 0.5), 5)
    tex = sum(a * b for a, b in sizes)
    out["mipchain_bc3_8192"] = {"ms": round(dt * 1e3, 3), "Mtexels_s": round(tex / dt / 1e6, 1), "algorithmic_GBs": round(tex * 5.0 / dt / 1e9, 1)}
    # cfg5 (1024 x 2048^2 RGBA8 -> BC7) on a bounded sample: 8 distinct 2048^2 images through the array entry point
    n5, side5 = 8, 2048
    imgs5 = [torch.roll(src.reshape(HEIGHT, WIDTH, 4)[:side5, :side5], shifts=(17 * i, 29 * i), dims=(0, 1)).contiguous() for i in range(n5)]
    sp5 = dx.compute_pitch(dx.DXGI_FORMAT_BC7_UNORM, side5, side5)[1]
    outs5 = [torch.empty(sp5, dtype=torch.uint8, device=dev) for _ in range(n5)]
    s5 = [dx.capi.device_image(t.data_ptr(), side5, side5, RGBA8) for t in imgs5]
    d5 = [dx.capi.device_image(t.data_ptr(), side5, side5, dx.DXGI_FORMAT_BC7_UNORM) for t in outs5]
    dt = timed(lambda: ctx.compress_many_device(s5, d5, 0, 0.5), 1)
    out["bc7_batch_8x2048"] = {"ms": round(dt * 1e3, 2), "Mtexels_s": round(n5 * side5 * side5 / dt / 1e6, 2), "sample": "8 of cfg5's 1024 images"}
    # The host-buffer boundary (dxtex_compress: pageable H2D + kernels + D2H), i.e. the PCIe-inclusive rate of the headline
    # and of BC1 -- never the metric, reported so the DESIGN.md note has a measured number behind it.
    for name, fmt, flags, n in (("bc7", dx.DXGI_FORMAT_BC7_UNORM, 0, 1), ("bc1", dx.DXGI_FORMAT_BC1_UNORM, 0, 5)):
        fn = lambda: ctx.compress(img, WIDTH, HEIGHT, RGBA8, fmt, flags, 0.5)
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        dt = (time.perf_counter() - t0) / n
        out[f"{name}_4096_host_buffers"] = {"ms": round(dt * 1e3, 2), "Mtexels_s": round(WIDTH * HEIGHT / dt / 1e6, 1)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads (BC1/BC3/BC6H/mips/decode) reported next to the headline")
    args = ap.parse_args()

    import torch
    import directxtex_amd as dx

    from directxtex_amd import sharding
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    rank, world = sharding.init_from_env("nccl", torch.device("cuda", local_rank))      # "nccl" is RCCL on ROCm
    distributed = world > 1
    n_gpus = world if distributed else 1
    if args.gpus != n_gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={n_gpus}", file=sys.stderr)

    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    ctx = dx.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)

    img = make_image(seed=2 + rank)                       # each GPU compresses its own image
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
    ctx.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    kernels = ctx.profile_end()

    # whole-job throughput: texels of all ranks / slowest rank's time (no data-path collective anywhere)
    elapsed, texels = sharding.aggregate(elapsed, float(WIDTH) * HEIGHT * args.steps, world, dev)
    value = texels / elapsed / 1e6

    if rank == 0:
        # ---- roofline of the dominant kernel -----------------------------------------------------------
        per_launch = {k: (ms / max(1, n)) for k, (ms, n) in kernels.items()}
        dom = max(per_launch, key=per_launch.get) if per_launch else None
        algo_bytes = ALGO_BYTES_PER_TEXEL * WIDTH * HEIGHT
        roof = None
        if dom:
            achieved = algo_bytes / (per_launch[dom] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                    "algorithmic_bytes_per_launch": int(algo_bytes), "kernel_ms": round(per_launch[dom], 4),
                    "note": "BC7 at the reference's search depth is VALU-bound (integer endpoint search); HBM "
                            "fraction is reported as the contract asks, VALU utilisation is in profiles/",
                    "all_kernels_ms": {k: round(v, 4) for k, v in sorted(per_launch.items(), key=lambda kv: -kv[1])},
                    "step_kernel_ms": round(sum(per_launch.values()), 4)}
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if roof and os.path.exists(pmc):
            try:
                t = json.load(open(pmc))
                if t.get("kernel") == dom:
                    roof["traffic"] = t.get("hbm_bytes_per_launch")
                    roof["traffic_source"] = t.get("source")
                    if t.get("valu"):
                        roof["valu_from_profile"] = t["valu"]      # SQ counters of the same kernel (profiles/): what actually bounds it
            except Exception:
                pass

        # ---- quality + CPU baseline (N = 1 only) ---------------------------------------------------------
        cpu = None
        extra = {}
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                res = cpu_baseline(img)
                if res:
                    cpu, (x0, y0, side, ref_payload) = res
                    import oracle
                    out = dst.cpu().numpy().reshape(HEIGHT // 4, WIDTH // 4, 16)
                    got = np.ascontiguousarray(out[y0 // 4:(y0 + side) // 4, x0 // 4:(x0 + side) // 4]).reshape(-1, 16)
                    ref = ref_payload.reshape(-1, 16)
                    crop = np.ascontiguousarray(img[y0:y0 + side, x0:x0 + side])
                    srcf = oracle.load_image(crop, side, side, 28)
                    extra["gpu_psnr_db_on_sample"] = round(oracle.psnr_rgb(oracle.decode_image(got.reshape(-1), side, side, 98)[..., :3], srcf[..., :3]), 3)
                    extra["blocks_identical_to_reference_on_sample"] = float((got == ref).all(axis=1).mean())
            except Exception as e:                              # the baseline must never break the bench line
                extra["cpu_baseline_error"] = repr(e)

        if n_gpus == 1 and not args.no_extra:
            try:
                extra["other_workloads"] = other_workloads(ctx, dev, img)
            except Exception as e:
                extra["other_workloads_error"] = repr(e)

        line = {
            "metric": "Mtexels/s BC7 encode (4096^2 RGBA8, TEX_COMPRESS_DEFAULT)",
            "value": round(value, 3), "unit": "Mtexels/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "4096x4096 RGBA8 -> BC7_UNORM, TEX_COMPRESS_DEFAULT, one image per GPU, source and "
                                   "payload resident in HBM (dxtex_compress_device)",
                       "image": "directxtex_amd.synth.rgba8 hash-noise recipe, opaque, seed 2+rank",
                       "sharding": f"image-per-GPU x{n_gpus}, no data-path collective"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        line.update(extra)
        print(json.dumps(line), flush=True)

    ctx.close()
    if distributed:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
