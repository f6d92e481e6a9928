#!/bin/bash
# Run on the GPU box through gpurun: rocprofv3 kernel trace + stats of a workload command, then separate PMC passes (PMC passes never carry a
# runtime / sys trace): SQ instruction counters; SQ wait / activity split (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, all in
# quad-cycles) + GRBM_GUI_ACTIVE for the clock; the LDS pass (bank-conflict cycles against all LDS cycles, LDS issue stalls, VMEM / scalar
# activity); TCC FETCH_SIZE; TCC WRITE_SIZE (the two TCC counters do not fit one pass). Outputs under gpurun_out/$1.
# usage: tools/profile.sh <tag> <workload: bc7|others>
cd /tmp && export TMPDIR=/tmp
TAG=$1; WL=$2; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/prof_workloads.py $WL"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace --output-format csv -- $CMD 3 > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d $OUT -o pass1 --output-format csv -- $CMD 1 > $OUT/pass1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT -o pass2 --output-format csv -- $CMD 1 > $OUT/pass2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_UNALIGNED_STALL -d $OUT -o pass3 --output-format csv -- $CMD 1 > $OUT/pass3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch --output-format csv -- $CMD 1 > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o write --output-format csv -- $CMD 1 > $OUT/write.log 2>&1
ls $OUT | tr '\n' ' '
