mkdir -p gpurun_out/r3e
ACMIL_GA2_WAVES=8 python -m pytest tests/test_ga_gpu.py tests/test_ga_fuzz_gpu.py tests/test_full_size_gpu.py -q -m gpu -k "not range_guard_falls" > gpurun_out/r3e/tests8.log 2>&1; echo "tests(8 waves) rc=$?"; tail -3 gpurun_out/r3e/tests8.log
python tools/abl_clock.py w4 16 > gpurun_out/r3e/clk.log 2>&1
ACMIL_GA2_WAVES=8 python tools/abl_clock.py w8skew 16 >> gpurun_out/r3e/clk.log 2>&1
ACMIL_GA2_WAVES=8 python tools/abl_clock.py w8skew_b1 1 >> gpurun_out/r3e/clk.log 2>&1
ACMIL_GA2_WAVES=8 python tools/abl_clock.py w8skew_f16 16 float16 >> gpurun_out/r3e/clk.log 2>&1
grep ABLCLK gpurun_out/r3e/clk.log
