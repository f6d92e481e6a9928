#!/bin/bash
# DEVELOPMENT TOOL: see tools/bc7_prune_stats.cpp. usage: tools/run_bc7_prune_stats.sh [ntiles]
set -e
HERE=$(cd "$(dirname "$0")" && pwd); N=${1:-2000}
mkdir -p "$HERE/../build"
python3 - "$N" "$HERE/../build/prune_tiles.bin" <<'PY'
import sys, numpy as np
sys.path.insert(0, sys.argv[0] and __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(sys.argv[2])), ".."))
from directxtex_amd import synth
n = int(sys.argv[1]); img = synth.survey_rgba8(4096, 4096, 2, "opaque")
rng = np.random.default_rng(5); bx = rng.integers(0, 1024, n); by = rng.integers(0, 1024, n)
open(sys.argv[2], "wb").write(b"".join(np.ascontiguousarray(img[y*4:y*4+4, x*4:x*4+4]).tobytes() for x, y in zip(bx, by)))
PY
/opt/rocm/lib/llvm/bin/clang++ -x hip --cuda-host-only -std=c++17 -O2 -ffp-contract=off -fno-fast-math -w -I/opt/rocm/include \
  "$HERE/bc7_prune_stats.cpp" -o "$HERE/../build/bc7_prune_stats" -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
"$HERE/../build/bc7_prune_stats" "$HERE/../build/prune_tiles.bin" "$N"
