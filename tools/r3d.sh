mkdir -p gpurun_out/r3d
ACMIL_GA2_WAVES=8 python -m pytest tests/test_ga_gpu.py tests/test_ga_fuzz_gpu.py -x -q -m gpu > gpurun_out/r3d/tests8.log 2>&1; echo "tests(8 waves) rc=$?"; tail -3 gpurun_out/r3d/tests8.log
python tools/abl_clock.py w4 16 > gpurun_out/r3d/clk.log 2>&1
ACMIL_GA2_WAVES=8 python tools/abl_clock.py w8 16 >> gpurun_out/r3d/clk.log 2>&1
ACMIL_GA2_WAVES=8 python tools/abl_clock.py w8_b1 1 >> gpurun_out/r3d/clk.log 2>&1
ACMIL_GA2_WAVES=8 python tools/abl_clock.py w8_f16 16 float16 >> gpurun_out/r3d/clk.log 2>&1
grep ABLCLK gpurun_out/r3d/clk.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3d/bench_drv.json 2> gpurun_out/r3d/bench_drv.err; python -c "
import json; d=json.loads(open('gpurun_out/r3d/bench_drv.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['module_slides_per_s'], d['roofline']['us_per_launch'])"
