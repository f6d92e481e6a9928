set -x
O=gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest tests/test_nonfinite_gpu.py tests/test_zz_threads_gpu.py "tests/test_zz_huge_gpu.py::test_huge_image_identical_to_the_reference[huge_bc1_16384]" -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
cat gpurun_out/nonfinite_report.txt | tail -70
