"""Image-sharded multi-GPU execution: a texture array / atlas is split one image per GPU,
round-robin; units are independent, so there is NO data-path collective. The only communication is the timing protocol
of the benchmark: a barrier on both sides of the timed region and a MAX over the ranks' elapsed times (plus a SUM of
the texels each rank processed). Works with any torch.distributed backend: "nccl" (= RCCL over xGMI) on the GPUs,
"gloo" in the CPU tests.

The one place with a real exchange step is a SINGLE image split over the GPUs ("a single huge image may be split by
block-rows", no halo for a block codec): every rank encodes a stripe of block rows and the stripes are gathered where the caller
wants the payload (stripe_rows, gather_stripes: one all_gather of the BC stripes - RCCL over xGMI on the GPUs)."""
import os


def images_for_rank(n_images, world, rank):
    """Image i goes to GPU i mod world (DirectXTexCompress.cpp:794-833 treats array items independently)."""
    return list(range(rank, n_images, world))


def run_shard(n_images, world, rank, load, compress_many, batch=128, limit=None):
    """This rank's part of an image-sharded job: images i = rank (mod world) of n_images, `batch` at a time through
    compress_many(list of images) -> list of payloads (Context.compress_many: dxtex_compress_many stages each batch's chunks
    through pinned memory while the previous chunk is searched). load(i) -> image i. Returns {image index: payload}.
    No rank ever touches another rank's images or results."""
    mine = images_for_rank(n_images, world, rank)
    if limit is not None:
        mine = mine[:limit]
    out = {}
    for at in range(0, len(mine), batch):
        idx = mine[at:at + batch]
        for i, payload in zip(idx, compress_many([load(i) for i in idx])):
            out[i] = payload
    return out


def stripe_rows(block_rows, world, rank):
    """Block rows [r0, r1) of a single image that rank `rank` of `world` encodes: contiguous, the first block_rows % world ranks take one
    more (a block codec needs no halo: blocks are independent, DirectXTexCompress.cpp:257-281)."""
    base, rem = divmod(int(block_rows), int(world))
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


def gather_stripes(stripe, block_rows, row_bytes, world, rank):
    """The one data-path collective of the package: every rank's stripe of BC block rows (a uint8 tensor of (r1 - r0) * row_bytes bytes, on
    the GPU with "nccl", anywhere with "gloo") -> the whole payload, on every rank (all_gather; stripes are padded to the longest one, which
    differs by at most one block row). Returns a uint8 tensor of block_rows * row_bytes bytes on the stripe's device."""
    import torch
    if world <= 1:
        return stripe
    import torch.distributed as dist
    longest = stripe_rows(block_rows, world, 0)[1] * row_bytes
    dev = stripe.device
    send = torch.zeros(longest, dtype=torch.uint8, device=dev)
    send[:stripe.numel()] = stripe
    if dist.get_backend() == "gloo":
        send = send.cpu()                                  # gloo gathers host tensors
    got = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(got, send)
    parts = []
    for r in range(world):
        r0, r1 = stripe_rows(block_rows, world, r)
        parts.append(got[r][:(r1 - r0) * row_bytes])
    return torch.cat(parts).to(dev)


def gather_index(results, world):
    """Bookkeeping only (object all_gather of (index, sha256) pairs - no texture data crosses ranks): every rank learns which image
    indices were compressed where, and a digest of each payload. Returns a list over ranks of {index: hex digest}."""
    import hashlib
    mine = {int(i): hashlib.sha256(bytes(memoryview(p))).hexdigest() for i, p in results.items()}
    if world <= 1:
        return [mine]
    import torch.distributed as dist
    got = [None] * world
    dist.all_gather_object(got, mine)
    return got


def gather_objects(obj, world):
    """Bookkeeping only: every rank's small python object (indices, counts) -> list over ranks. No texture data."""
    if world <= 1:
        return [obj]
    import torch.distributed as dist
    got = [None] * world
    dist.all_gather_object(got, obj)
    return got


def init_from_env(backend, device=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* when WORLD_SIZE > 1. Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
            dist.init_process_group(backend, **kw)
    return rank, world


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def aggregate(elapsed_s, texels_local, world, device="cpu"):
    """-> (max elapsed over ranks, total texels over ranks): whole-job throughput = total / max."""
    if world <= 1:
        return float(elapsed_s), float(texels_local)
    import torch
    import torch.distributed as dist
    if dist.get_backend() == "gloo":
        device = "cpu"                                   # gloo reduces host tensors
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    n = torch.tensor([texels_local], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), float(n.item())
