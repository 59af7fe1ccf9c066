mkdir -p gpurun_out/r3h
ACMIL_GA2_PAIR=1 python -m pytest tests/test_ga_gpu.py tests/test_ga_fuzz_gpu.py tests/test_full_size_gpu.py -q -m gpu -x > gpurun_out/r3h/tests_pair.log 2>&1; echo "tests(pair) rc=$?"; tail -2 gpurun_out/r3h/tests_pair.log
: > gpurun_out/r3h/clk.log
for i in 1 2; do
python tools/abl_clock.py w4_$i 16 >> gpurun_out/r3h/clk.log 2>&1
ACMIL_GA2_PAIR=1 python tools/abl_clock.py pair4_$i 16 >> gpurun_out/r3h/clk.log 2>&1
done
ACMIL_GA2_PAIR=1 python tools/abl_clock.py pair4_b1 1 >> gpurun_out/r3h/clk.log 2>&1
python tools/abl_clock.py w4_b1 1 >> gpurun_out/r3h/clk.log 2>&1
grep ABLCLK gpurun_out/r3h/clk.log | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l.split('ABLCLK ')[1]); print('%-14s %7.1f us  %4d MHz  %4d W  %.3f Mcyc' % (d['name'], d['us_per_launch'], d['sclk_mhz_mean'], d['power_w_mean'], d['us_per_launch']*d['sclk_mhz_mean']/1e6))"
