"""bench.py's N > 1 path executed before the driver does: two ranks launched exactly as the driver launches them
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ...`), sharing GPU 0 through the gloo
backend (`--backend gloo`; LOCAL_RANK is taken modulo the device count). Everything except `init_process_group("nccl")` itself is the
code of the 8-GPU run: rank-dependent images, barriers, the MAX / SUM aggregation, the cfg5 shard with its disjoint index sets and
digest check, one JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.gpu
def test_two_ranks_one_gpu():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--backend", "gloo", "--no-cpu-baseline", "--cfg5-images", "10"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # ONE line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["scaling"] == "weak"
    # whole-job value = texels of both ranks / the slower rank's time
    texels = 2 * 4096 * 4096 * line["steps"]
    assert abs(line["value"] - texels / (line["ms_per_step"] * 1e-3 * line["steps"]) / 1e6) <= 0.01 * line["value"]
    # the single-image split (strong scaling): both ranks encode half of rank 0's image, the stripes are all_gathered, the payload is the reference's
    sp = line["other_workloads"]["bc7_4096_split"]
    assert "error" not in sp, sp
    assert sp["n_gpus"] == 2 and sp["scaling"] == "strong" and sp["ms"] > 0
    if os.path.exists(os.path.join(ROOT, "tests", "golden", "fullsize.json")):
        assert sp["identical_to_reference_golden"] is True, sp
    c5 = line["other_workloads"]["cfg5_shard"]
    assert "error" not in c5, c5
    assert c5["images"] == 20 and c5["indices_disjoint"] is True
    assert c5["indices_per_rank"][0][:2] == [0, 2] and c5["indices_per_rank"][1][:2] == [1, 3]         # image i on rank i mod N
    assert c5["chunks"] == 2
    if os.path.exists(os.path.join(ROOT, "tests", "golden", "cfg5.json")):
        assert c5["payloads_checked"] == 20 and c5["identical_to_reference_golden"] is True, c5
