python tools/r04_quick.py cfg4 2>&1 | grep cfg4
timeout 900 python -m pytest tests/test_scanline_parity.py tests/test_golden.py tests/test_host_api.py tests/test_dxtexconv.py -m gpu -q -x 2>&1 | tail -3
