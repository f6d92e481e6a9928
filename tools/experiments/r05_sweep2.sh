#!/bin/bash
# DEVELOPMENT: round-5 sweep 2: the fill loop of Exhaustive, the filtered PerturbOne for mode 4's 3-bit colours, and the old knobs again after the peel
O=gpurun_out/r05_sweep2; mkdir -p $O
run() { echo "=== $*"; env "$@" PROBE_TOP=${TOP:-0} timeout 120 python tools/quick_probe.py --dev bc7 2>&1 | grep -v amdgpu.ids; }
{
TOP=45 run X=0
for f in 32 40 48 56 64; do TOP=45 run DXTEX_BC7_FILL_BELOW=$f | grep -E "===|per image|payload|exhaustive"; done
for t in 32 40 56 64; do run DXTEX_BC7_TAIL_BELOW=$t; done
for r in 3 4 6 7 9; do run DXTEX_BC7_RANGE_TESTS=$r; done
run DXTEX_BC7_PEEL_LAYERS=2 DXTEX_BC7_RANGE_TESTS=9
for e in 60 80 120 150; do run DXTEX_BC7_EARLY6_PCT=$e; done
for e in 10 40 60; do run DXTEX_BC7_EARLY6_MIN_PCT=$e; done
run DXTEX_BC7_SERIAL=1
PROBE_TOP=45 bash tools/ab_variants.sh run "python tools/quick_probe.py --dev bc7" | grep -E "===|per image|payload|perturb_mode4_im1"
} > $O/log.txt 2>&1
grep -E "===|per image|DIFFERS" $O/log.txt
