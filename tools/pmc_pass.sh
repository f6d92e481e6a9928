#!/bin/bash
# DEVELOPMENT TOOL (run on the GPU box through gpurun): separate rocprofv3 --pmc passes over tools/prof_bc7.py.
# usage: tools/pmc_pass.sh <outdir-under-gpurun_out> <size> "<counters pass 1>" "<counters pass 2>" ...
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; SIZE=$2; shift 2
mkdir -p $OUT
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT -o pass$i --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_bc7.py $SIZE > $OUT/pass$i.log 2>&1
  tail -2 $OUT/pass$i.log
done
ls $OUT
