"""The N > 1 path of bench.py on CPU: two processes, gloo backend, same sharding and timing-aggregation code
(directxtex_amd/sharding.py) the GPU run uses with nccl/RCCL. No collective touches texture data."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, time, json
sys.path.insert(0, %r)
from directxtex_amd import sharding
rank, world = sharding.init_from_env("gloo")
mine = sharding.images_for_rank(11, world, rank)
sharding.barrier(world)
t0 = time.perf_counter()
time.sleep(0.05 * (rank + 1))                     # rank 1 is the slow one
elapsed = time.perf_counter() - t0
sharding.barrier(world)
tmax, total = sharding.aggregate(elapsed, len(mine) * 2048 * 2048, world)
print(json.dumps({"rank": rank, "world": world, "mine": mine, "tmax": tmax, "total": total, "elapsed": elapsed}), flush=True)
import torch.distributed as dist
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_rank_sharding_and_timing():
    import json
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("{")][-1]))
    outs.sort(key=lambda d: d["rank"])
    # every image exactly once, round-robin
    assert outs[0]["mine"] == [0, 2, 4, 6, 8, 10] and outs[1]["mine"] == [1, 3, 5, 7, 9]
    # both ranks agree on the aggregate: MAX of the times (the slow rank's), SUM of the texels
    assert outs[0]["tmax"] == outs[1]["tmax"] >= outs[1]["elapsed"] - 1e-9
    assert outs[0]["tmax"] >= max(o["elapsed"] for o in outs) - 1e-9
    assert outs[0]["total"] == outs[1]["total"] == 11 * 2048 * 2048


SHARD_WORKER = r'''
import os, sys, json, hashlib
import numpy as np
sys.path.insert(0, %r)
from directxtex_amd import sharding
rank, world = sharding.init_from_env("gloo")
N = 37
calls = []
def load(i):                                   # image i: 8 x 8 RGBA8 that depends on i only
    return (np.arange(256, dtype=np.uint32).reshape(8, 8, 4) * (i + 1) %% 251).astype(np.uint8)
def compress_many(imgs):                       # stand-in for Context.compress_many (no GPU here): one "payload" per image, batch-independent
    calls.append(len(imgs))
    return [np.frombuffer(hashlib.sha256(im.tobytes()).digest(), np.uint8) for im in imgs]
res = sharding.run_shard(N, world, rank, load, compress_many, batch=5)
index = sharding.gather_index(res, world)
print(json.dumps({"rank": rank, "keys": sorted(res), "calls": calls, "index": [{str(k): v for k, v in d.items()} for d in index]}), flush=True)
import torch.distributed as dist
dist.destroy_process_group()
'''


def test_two_rank_shard_covers_every_image_once():
    """cfg5's driver (sharding.run_shard, what bench.py's cfg5 leg and a batch tool call): with two ranks every image index is
    compressed exactly once, on the rank i mod 2, in batches, and the assembled result does not depend on who did what - it equals
    the single-process result."""
    import hashlib
    import json
    import numpy as np
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", SHARD_WORKER % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("{")][-1]))
    outs.sort(key=lambda d: d["rank"])
    assert outs[0]["keys"] == list(range(0, 37, 2)) and outs[1]["keys"] == list(range(1, 37, 2))
    assert outs[0]["calls"] == [5, 5, 5, 4] and outs[1]["calls"] == [5, 5, 5, 3]          # batches of 5, the last one short
    # both ranks hold the same bookkeeping; the union has every index exactly once
    assert outs[0]["index"] == outs[1]["index"]
    merged = {}
    for d in outs[0]["index"]:
        for k, v in d.items():
            assert k not in merged, f"image {k} compressed twice"
            merged[k] = v
    assert sorted(int(k) for k in merged) == list(range(37))
    # ... and equals what one process computes for the whole array
    sys.path.insert(0, ROOT)
    from directxtex_amd import sharding

    def load(i):
        return (np.arange(256, dtype=np.uint32).reshape(8, 8, 4) * (i + 1) % 251).astype(np.uint8)
    single = sharding.run_shard(37, 1, 0, load, lambda imgs: [np.frombuffer(hashlib.sha256(im.tobytes()).digest(), np.uint8) for im in imgs], batch=37)
    assert {str(i): hashlib.sha256(p.tobytes()).hexdigest() for i, p in single.items()} == merged


def test_single_rank_is_identity():
    sys.path.insert(0, ROOT)
    from directxtex_amd import sharding
    assert sharding.images_for_rank(5, 1, 0) == [0, 1, 2, 3, 4]
    assert sharding.aggregate(0.5, 100, 1) == (0.5, 100.0)


STRIPE_WORKER = r'''
import os, sys, json, hashlib
import numpy as np, torch
sys.path.insert(0, %r)
from directxtex_amd import sharding
rank, world = sharding.init_from_env("gloo")
rows, row_bytes = 11, 48                              # 11 block rows over 3 ranks: 4 + 4 + 3
whole = (np.arange(rows * row_bytes, dtype=np.uint32) * 2654435761 >> 13).astype(np.uint8)
r0, r1 = sharding.stripe_rows(rows, world, rank)
mine = torch.from_numpy(whole[r0 * row_bytes:r1 * row_bytes].copy())
got = sharding.gather_stripes(mine, rows, row_bytes, world, rank)
print(json.dumps({"rank": rank, "r": [r0, r1], "same": bool(np.array_equal(got.numpy(), whole)), "n": int(got.numel())}), flush=True)
import torch.distributed as dist
dist.destroy_process_group()
'''


def test_one_image_split_by_block_rows():
    """The single-image split: stripes of block rows are contiguous, cover the image exactly once whatever the rank count, and
    the all_gather of the (padded) stripes reassembles the payload on every rank - three gloo ranks, eleven block rows."""
    import json
    from directxtex_amd import sharding
    for rows in (1, 7, 8, 1024, 1026):
        for world in (1, 2, 3, 8):
            cover = [sharding.stripe_rows(rows, world, r) for r in range(world)]
            assert cover[0][0] == 0 and cover[-1][1] == rows
            assert all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
            sizes = [b - a for a, b in cover]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    port = _free_port()
    procs = []
    for rank in range(3):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", STRIPE_WORKER % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("{")][-1]))
    outs.sort(key=lambda d: d["rank"])
    assert [o["r"] for o in outs] == [[0, 4], [4, 8], [8, 11]]
    assert all(o["same"] and o["n"] == 11 * 48 for o in outs)
