#!/bin/bash
# DEVELOPMENT: round-5 lone-image sweep: the early mode-6 threshold and the small plan at 256^2 ... 2048^2
run() { echo "=== $*"; env "$@" PROBE_TOP=0 PROBE_SIZES=256,512,1024,1448,2048 timeout 120 python tools/quick_probe.py --dev small 2>&1 | grep "per image"; }
run X=0
run DXTEX_BC7_EARLY6_MIN_PCT=25
run DXTEX_BC7_EARLY6_MIN_PCT=0
run DXTEX_BC7_NO_SMALL_PLAN=1
run DXTEX_BC7_NO_SMALL_PLAN=1 DXTEX_BC7_EARLY6_MIN_PCT=25
run DXTEX_BC7_SMALL_PLAN="1,26/3,2/16,14,15,18,7,0|24/28/25" DXTEX_BC7_EARLY6_MIN_PCT=25
run DXTEX_BC7_SMALL_PLAN="1/3,2/16,14,15,18,7,0|24/28/25|26" DXTEX_BC7_EARLY6_MIN_PCT=101
run DXTEX_BC7_SMALL_PLAN="1/3,2/26,14,15,18,7,0|24/28/25" DXTEX_BC7_EARLY6_MIN_PCT=101
