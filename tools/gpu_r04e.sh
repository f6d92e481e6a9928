O=gpurun_out/r04e; mkdir -p $O
python tools/r04_quick.py cfg4 bc1 2>&1 | grep "cfg4\|4096^2"
timeout 900 python -m pytest tests/test_scanline_parity.py tests/test_golden.py tests/test_bc15_parity.py tests/test_host_api.py tests/test_dxtexconv.py tests/test_bc7_parity.py tests/test_many_gpu.py "tests/test_zz_fullsize_gpu.py" -m gpu -q -x -k "not live" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
