"""Image-sharded multi-GPU execution: a texture array / atlas is split one image per GPU,
round-robin; units are independent, so there is NO data-path collective. The only communication is the timing protocol
of the benchmark: a barrier on both sides of the timed region and a MAX over the ranks' elapsed times (plus a SUM of
the texels each rank processed). Works with any torch.distributed backend: "nccl" (= RCCL over xGMI) on the GPUs,
"gloo" in the CPU tests."""
import os


def images_for_rank(n_images, world, rank):
    """Image i goes to GPU i mod world (DirectXTexCompress.cpp:794-833 treats array items independently)."""
    return list(range(rank, n_images, world))


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
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    n = torch.tensor([texels_local], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), float(n.item())
