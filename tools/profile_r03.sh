#!/bin/bash
# Run on the GPU box through gpurun: rocprofv3 kernel trace + stats of a workload command, then separate PMC passes
# (SQ instruction / issue counters; GRBM_GUI_ACTIVE for the effective clock; TCC FETCH_SIZE; TCC WRITE_SIZE - the TCC counters do not
# fit one pass, and PMC passes never carry a runtime/sys trace). Outputs under gpurun_out/$1.
# usage: tools/profile_r03.sh <tag> <workload: bc7|others>
cd /tmp && export TMPDIR=/tmp
TAG=$1; WL=$2; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/prof_workloads.py $WL"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace --output-format csv -- $CMD 3 > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d $OUT -o pass1 --output-format csv -- $CMD 1 > $OUT/pass1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d $OUT -o pass2 --output-format csv -- $CMD 1 > $OUT/pass2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch --output-format csv -- $CMD 1 > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o write --output-format csv -- $CMD 1 > $OUT/write.log 2>&1
ls $OUT | tr '\n' ' '
