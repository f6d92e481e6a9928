#!/bin/bash
# Per-kernel register / scratch / LDS / occupancy table of one .hip file (compiler remarks, no GPU needed).
# usage: tools/kernel_resources.sh directxtex_amd/csrc/bc7_encode.hip
HERE=$(cd "$(dirname "$0")" && pwd)
SRC=$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
  -I"$HERE/../directxtex_amd/csrc" -I"$HERE/../include" -x hip -c "$SRC" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys, re, subprocess
rows = []; cur = None
for line in sys.stdin:
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        try: name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        except Exception: pass
        name = name.replace("dxtex::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        cur = {"name": name}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
print("| kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | LDS B | occupancy waves/SIMD |")
print("|---|---|---|---|---|---|---|")
for r in rows:
    print("| `%s` | %s | %s | %s | %s | %s | %s |" % (r["name"], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
'
