#!/bin/bash
# DEVELOPMENT TOOL (run on the GPU box through gpurun): two rocprofv3 --pmc passes (SQ instruction counters; SQ wait / LDS counters) over the cfg3
# BC6H encode of tools/quick_probe.py - what the round-4 work on bc6h_perturb_filter_kernel was steered by. Outputs under gpurun_out/f6pmc.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/f6pmc; mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/quick_probe.py bc6h"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT -o p1 --output-format csv -- $CMD > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD -d $OUT -o p2 --output-format csv -- $CMD > $OUT/p2.log 2>&1
ls $OUT
