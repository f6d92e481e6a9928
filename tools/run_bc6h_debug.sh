#!/bin/bash
# DEVELOPMENT TOOL: host-side comparison of the BC6H core against the reference (see bc6h_debug.cpp).
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
mkdir -p "$HERE/../build"
/opt/rocm/lib/llvm/bin/clang++ -x hip --cuda-host-only -std=c++17 -O1 -ffp-contract=off -fno-fast-math -w $DXTEX_DEBUG_DEFS \
  -I"$HERE/../oracle/shim" -I/root/reference/DirectXTex -I/opt/rocm/include \
  "$HERE/bc6h_debug.cpp" -o "$HERE/../build/bc6h_debug" -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
"$HERE/../build/bc6h_debug" "$@"
