mkdir -p gpurun_out/r3f
ACMIL_GA2_PAIR=1 python -m pytest tests/test_ga_gpu.py tests/test_ga_fuzz_gpu.py tests/test_full_size_gpu.py tests/test_train_gpu.py -q -m gpu -x > gpurun_out/r3f/tests_pair.log 2>&1; echo "tests(pair) rc=$?"; tail -5 gpurun_out/r3f/tests_pair.log
python tools/abl_clock.py w4 16 > gpurun_out/r3f/clk.log 2>&1
ACMIL_GA2_PAIR=1 python tools/abl_clock.py pair4 16 >> gpurun_out/r3f/clk.log 2>&1
ACMIL_GA2_PAIR=1 ACMIL_GA2_WAVES=8 python tools/abl_clock.py pair8 16 >> gpurun_out/r3f/clk.log 2>&1
ACMIL_GA2_PAIR=1 python tools/abl_clock.py pair4_b1 1 >> gpurun_out/r3f/clk.log 2>&1
ACMIL_GA2_PAIR=1 python tools/abl_clock.py pair4_f16 16 float16 >> gpurun_out/r3f/clk.log 2>&1
grep ABLCLK gpurun_out/r3f/clk.log
