#!/bin/bash
# DEVELOPMENT TOOL: builds variants of the library that differ in compile-time definitions of ONE translation unit, for A/B runs on
# the GPU box in one gpurun call.
#   build (here):   tools/ab_variants.sh build bc7_encode.hip  v6="-DDXTEX_ROUGH_WGS=6" v7="-DDXTEX_ROUGH_WGS=7" ...
#   run (GPU box):  tools/ab_variants.sh run "python tools/quick_probe.py --dev bc7" -> runs the command once per variant in build/variants/
# A variant replaces lib/libdxtex_amd_dev.so for its run (the command selects the development build itself: --dev); the product library is not touched.
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$HERE/..
CS=$ROOT/directxtex_amd/csrc; OBJ=$ROOT/build/obj; VAR=${VARDIR:-$ROOT/build/variants}      # VARDIR: several sets of variants side by side
if [ "$1" = build ]; then
  TU=$2; shift 2
  rm -rf $VAR; mkdir -p $VAR
  OTHERS=$(ls $OBJ/*.hip.o $OBJ/*.cpp.o | grep -v "/$TU.o" | grep -v "/dev_")
  for spec in "$@"; do
    name=${spec%%=*}; defs=${spec#*=}
    /opt/rocm/bin/hipcc $defs --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -DDXTEX_DEV \
      -I$CS -I$ROOT/include -x hip -c $CS/$TU -o $VAR/$name.o &
  done
  wait
  for spec in "$@"; do
    name=${spec%%=*}
    # capi.cpp / the other knob-reading units come from the dev objects where they exist
    LINK=""
    for o in $OTHERS; do b=$(basename $o); if [ -f $OBJ/dev_${b%.o}.o ] ; then LINK="$LINK $OBJ/dev_${b%.o}.o"; else LINK="$LINK $o"; fi; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $VAR/$name.so $LINK $VAR/$name.o -Wl,-soname,libdxtex_amd_dev.so -Wl,-Bsymbolic-functions -Wl,-rpath,/opt/rocm/lib
    rm $VAR/$name.o
  done
  ls -la $VAR
else
  shift
  cp $ROOT/directxtex_amd/lib/libdxtex_amd_dev.so /tmp/dev_keep.so
  for v in $VAR/*.so; do
    echo "=== variant $(basename $v .so)"
    cp $v $ROOT/directxtex_amd/lib/libdxtex_amd_dev.so
    bash -c "$*" 2>&1 | grep -v amdgpu.ids
  done
  cp /tmp/dev_keep.so $ROOT/directxtex_amd/lib/libdxtex_amd_dev.so
fi
