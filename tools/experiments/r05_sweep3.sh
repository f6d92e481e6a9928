#!/bin/bash
# DEVELOPMENT: round-5 sweep 3: mode orders and the early mode-6 phase after the peel; occupancy variants of the mode 4 / 5 kernels
O=gpurun_out/r05_sweep3; mkdir -p $O
run() { echo "=== $*"; env "$@" PROBE_TOP=${TOP:-0} timeout 120 python tools/quick_probe.py --dev bc7 2>&1 | grep -v amdgpu.ids; }
{
run X=0; run X=1; run X=2
run DXTEX_BC7_EARLY6_MIN_PCT=101; run DXTEX_BC7_EARLY6_MIN_PCT=101; run DXTEX_BC7_EARLY6_MIN_PCT=101
run DXTEX_BC7_ORDER=16,7,14,18,15,0,1,2,24,28,25,3,26
run DXTEX_BC7_ORDER=16,7,14,18,15,0,1,2,24,28,25,26,3
run DXTEX_BC7_ORDER=16,7,14,18,15,0,3,2,1,24,28,25,26
run DXTEX_BC7_ORDER=7,14,18,15,0,1,16,2,3,24,28,25,26
run DXTEX_BC7_ORDER=7,14,18,15,0,1,2,3,16,24,28,25,26
run DXTEX_BC7_ORDER=16,7,14,18,15,0,1,2,3,26,24,28,25
run DXTEX_BC7_ORDER=16,7,14,18,15,0,1,2,3,25,24,28,26
run DXTEX_BC7_EARLY6_MIN_PCT=101 DXTEX_BC7_ORDER=16,7,14,18,15,0,1,26,2,3,24,28,25
run DXTEX_BC7_EARLY6_MIN_PCT=101 DXTEX_BC7_ORDER=16,7,14,18,15,0,1,2,3,26,24,28,25
PROBE_TOP=0 bash tools/ab_variants.sh run "python tools/quick_probe.py --dev bc7"
} > $O/log.txt 2>&1
grep -E "===|per image|DIFFERS" $O/log.txt
