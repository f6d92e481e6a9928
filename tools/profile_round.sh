#!/bin/bash
# Run on the GPU box through gpurun: rocprofv3 kernel trace + stats of the benchmark command, then separate PMC passes
# (SQ utilisation; TCC FETCH_SIZE; TCC WRITE_SIZE - the two TCC counters do not fit one pass). Outputs under gpurun_out/$1.
cd /tmp && export TMPDIR=/tmp
TAG=${1:-prof}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT -o trace --output-format csv -- $BENCH > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log | cut -c1-300
BENCH1="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d $OUT -o pass1 --output-format csv -- $BENCH1 > $OUT/pass1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH -d $OUT -o pass2 --output-format csv -- $BENCH1 > $OUT/pass2.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch --output-format csv -- $BENCH1 > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o write --output-format csv -- $BENCH1 > $OUT/write.log 2>&1
ls $OUT | tr '\n' ' '
