O=gpurun_out/r04c; mkdir -p $O
bash tools/ab_variants.sh run "python tools/r04_quick.py --dev bc1 2>&1 | grep -v '^  bc\|serial'" > $O/ab_bc15.txt 2>&1
python tools/r04_quick.py cfg4 > $O/cfg4.txt 2>&1
timeout 600 python -m pytest tests/test_scanline_parity.py tests/test_bc15_parity.py tests/test_golden.py "tests/test_zz_huge_gpu.py" -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
cat $O/ab_bc15.txt $O/cfg4.txt; tail -4 $O/pytest.log
