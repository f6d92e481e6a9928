O=$GRAFT_REPO_ROOT/gpurun_out/r04i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O -o small --output-format csv -- python $GRAFT_REPO_ROOT/tools/small_probe.py 512 3 > $O/small.log 2>&1
DXTEX_BC7_NO_SMALL_PLAN=1 timeout 300 rocprofv3 --kernel-trace -d $O -o serial --output-format csv -- python $GRAFT_REPO_ROOT/tools/small_probe.py --dev 512 3 > $O/serial.log 2>&1
ls $O
