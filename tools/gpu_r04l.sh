O=gpurun_out/r04l; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
