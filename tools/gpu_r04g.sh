python tools/r04_quick.py bc6h 2>&1 | grep -v "^  bc6h_\(bin\|store\)" | head -24
timeout 900 python -m pytest tests/test_bc6h_parity.py tests/test_golden.py tests/test_nonfinite_gpu.py tests/test_compress_formats.py -m gpu -q -x -k "bc6h or BC6H or nonfinite or golden or 95 or 96" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_bc7_parity.py -m gpu -q -x -k "pruning or array" 2>&1 | tail -2
