#!/bin/bash
# DEVELOPMENT: round-5 sweep 1 on the GPU box (one gpurun call): wave caps and staggered passes of the BC7 pipeline, development library.
O=gpurun_out/r05_sweep1; mkdir -p $O
P="python tools/quick_probe.py --dev bc7"
run() { echo "=== $*"; env "$@" PROBE_TOP=${TOP:-0} timeout 120 $P 2>&1 | grep -v amdgpu.ids; }
{
TOP=45 run X=0
python tools/quick_probe.py bc1 2>&1 | grep -E "ms per image|bc15_encode" | grep -v amdgpu.ids
for w in 4096 3072 2048; do run DXTEX_BC7_SEARCH_WAVES=$w; done
for w in 2560 1536 1024; do run DXTEX_BC7_FORK_WAVES=$w; done
for w in 1536 2048 3072 8192; do run DXTEX_BC7_STAGGER=2 DXTEX_BC7_STAGGER_WAVES=$w; done
for a in -1 3 16; do run DXTEX_BC7_STAGGER=2 DXTEX_BC7_STAGGER_AFTER=$a; done
for f in 0 2; do run DXTEX_BC7_STAGGER=2 DXTEX_BC7_STAGGER_FORK=$f; done
for p in 3 4; do run DXTEX_BC7_STAGGER=$p; run DXTEX_BC7_STAGGER=$p DXTEX_BC7_STAGGER_WAVES=1536;  done
run DXTEX_BC7_STAGGER=4 DXTEX_BC7_STAGGER_MIN=100000 DXTEX_BC7_SMALL_BLOCKS=100000
} > $O/log.txt 2>&1
tail -c 6000 $O/log.txt
