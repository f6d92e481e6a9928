#!/bin/bash
# DEVELOPMENT TOOL: host-side stage-by-stage comparison of the BC7 core against the reference (see bc7_debug.cpp).
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
mkdir -p "$HERE/../build"
/opt/rocm/lib/llvm/bin/clang++ -x hip --cuda-host-only -std=c++17 -O1 -ffp-contract=off -fno-fast-math -w $DXTEX_DEBUG_DEFS \
  -I"$HERE/../oracle/shim" -I/root/reference/DirectXTex -I/opt/rocm/include \
  "$HERE/bc7_debug.cpp" -o "$HERE/../build/bc7_debug" -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
"$HERE/../build/bc7_debug" "$@"
