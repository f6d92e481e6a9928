"""north_star's in-process shape of the 8-GPU batch: "one context per GPU used concurrently from 8 host threads" (threading
row; DirectXTexCompress.cpp:794-833 is the array loop the threads share out). Eight host threads, eight contexts (context i on GPU
i mod device_count: on a one-GPU box they share it), every thread pushing its own array through dxtex_compress_many - pinned staging,
copy streams, the BC7 search scratch, the side streams of modes 4 / 5 all exist eight times at once - while the other seven do the same.
Contexts share nothing, so every payload must be what a lone context produces: the small images are compared with the reference run
here, the 2048^2 cfg5 images with the reference's committed digests (tests/golden/cfg5.json)."""
import hashlib
import json
import os
import threading

import numpy as np
import pytest

import directxtex_amd as dx
from directxtex_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RGBA8, RGBA16F, BC1, BC3, BC6H, BC7 = 28, 10, 71, 77, 95, 98
NTHREADS = 8


def test_eight_threads_eight_contexts(oracle):
    import torch
    ndev = max(1, torch.cuda.device_count())
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg5.json")))["images"]
    # the distinct small images (sizes with and without partial blocks) and their reference payloads, computed once
    small = []
    for k, (w, h) in enumerate([(96, 64), (50, 34), (128, 128), (17, 9), (64, 200), (4, 4)]):
        img = synth.rgba8(w, h, seed=300 + k, alpha=("smooth", "opaque", "binary")[k % 3])
        small.append((img, w, h))
    want = {}
    for fmt in (BC7, BC1, BC3):
        want[fmt] = [oracle.ref_compress_image(img, w, h, RGBA8, fmt, 0, 0.5) for img, w, h in small]
    hdr = [((img.astype(np.float32) / 255 * 6).astype(np.float16), w, h) for img, w, h in small[:4]]
    want[BC6H] = [oracle.ref_compress_image(p, w, h, RGBA16F, BC6H, 0, 0.5) for p, w, h in hdr]
    big = {i: synth.survey_rgba8(2048, 2048, 1000 + i, "opaque") for i in range(NTHREADS)}
    for i, im in big.items():
        assert hashlib.sha256(im.tobytes()).hexdigest() == gold[str(i)]["input_sha256"]

    errors = []
    start = threading.Barrier(NTHREADS)

    def worker(t):
        try:
            ctx = dx.Context(t % ndev)
            start.wait()
            for rnd in range(2):
                order = [(i * (t + 1) + rnd) % len(small) for i in range(len(small))]          # every thread its own order of sizes
                for fmt in (BC7, BC1, BC3):
                    got = ctx.compress_array([(small[i][0], small[i][1], small[i][2], RGBA8, None) for i in order], fmt, 0, 0.5)
                    for i, g in zip(order, got):
                        if not np.array_equal(g, want[fmt][i]):
                            errors.append(f"thread {t} round {rnd}: format {fmt} image {i} differs from the reference")
                got = ctx.compress_array([(p, w, h, RGBA16F, None) for p, w, h in hdr], BC6H, 0, 0.5)
                for i, g in enumerate(got):
                    if not np.array_equal(g, want[BC6H][i]):
                        errors.append(f"thread {t} round {rnd}: BC6H image {i} differs from the reference")
                # a full cfg5 image per thread: eight 2048^2 BC7 searches (each with its own ~0.6 GiB of scratch) at once
                pay = ctx.compress_many([big[t]], 2048, 2048, RGBA8, BC7, 0, 0.5)[0]
                if hashlib.sha256(pay.tobytes()).hexdigest() != gold[str(t)]["sha256"]:
                    errors.append(f"thread {t} round {rnd}: cfg5 image {t} differs from the reference's digest")
            ctx.close()
        except Exception as e:      # noqa: BLE001 - reported below, a thread must not die silently
            errors.append(f"thread {t}: {e!r}")
            try:
                start.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(NTHREADS)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=900)
    assert not any(th.is_alive() for th in threads), "a worker thread did not finish"
    assert not errors, "\n".join(errors[:10])
