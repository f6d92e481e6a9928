#!/bin/bash
O=gpurun_out/r05_wt; mkdir -p $O
DXTEX_BC7_SERIAL=1 PROBE_TOP=0 PROBE_REPS=1 bash tools/ab_variants.sh run "python tools/quick_probe.py --dev bc7" > $O/serial.txt 2>&1
PROBE_TOP=0 PROBE_REPS=1 bash tools/ab_variants.sh run "python tools/quick_probe.py --dev bc7" > $O/fork.txt 2>&1
grep -E "^wt|per image" $O/serial.txt | cut -c1-900
