#!/bin/bash
# DEVELOPMENT TOOL (GPU box, through gpurun): per-dispatch kernel trace of the 8192^2 mip chains (box, cubic) -> gpurun_out/mips_trace.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mt && timeout 300 rocprofv3 --kernel-trace -d /tmp/mt -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/r03_probe.py mips > /tmp/mt.log 2>&1
python - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/mips_trace.txt
import csv, glob, collections
f = glob.glob('/tmp/mt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# last occurrence of each (kernel, grid) in order: print the final 60 dispatches
out = []
for r in [r for r in rows if 'resize' in r['Kernel_Name']][-60:]:
    n = r['Kernel_Name'].replace('dxtex::(anonymous namespace)::', '').split('(')[0]
    out.append("%-44s grid %8s x %5s  %8.2f us" % (n, r['Grid_Size_X'], r['Grid_Size_Y'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
print("\n".join(out))
PY
tail -45 $GRAFT_REPO_ROOT/gpurun_out/mips_trace.txt
