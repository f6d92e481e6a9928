set -x
O=gpurun_out/r04a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_zz_fullsize_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 600 $O/bench.err
timeout 900 bash tools/profile_r04.sh r04a_bc7 bc7
timeout 1200 bash tools/profile_r04.sh r04a_others others
