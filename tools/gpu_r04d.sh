O=$GRAFT_REPO_ROOT/gpurun_out/r04d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cp $GRAFT_REPO_ROOT/directxtex_amd/lib/libdxtex_amd_dev.so /tmp/dev_keep.so
for v in base nostream over2 nobc45; do
  cp $GRAFT_REPO_ROOT/build/variants/$v.so $GRAFT_REPO_ROOT/directxtex_amd/lib/libdxtex_amd_dev.so
  timeout 300 rocprofv3 --kernel-trace --stats -d $O -o kt_$v --output-format csv -- python $GRAFT_REPO_ROOT/tools/r04_quick.py --dev bc1 > $O/kt_$v.log 2>&1
  echo "== $v"; grep -i "bc15" $O/kt_${v}_kernel_stats.csv | cut -c1-60,150-240; grep sha256 $O/kt_$v.log | tr '\n' ' '; echo
done
cp /tmp/dev_keep.so $GRAFT_REPO_ROOT/directxtex_amd/lib/libdxtex_amd_dev.so
cd $GRAFT_REPO_ROOT; timeout 600 python -m pytest tests/test_bc7_parity.py -m gpu -q -x -k "multi or array or pruning" 2>&1 | tail -3
