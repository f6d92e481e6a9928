timeout 1200 python -m pytest tests/test_bench_ranks_gpu.py -m gpu -q -x 2>&1 | tail -15
