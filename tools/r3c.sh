mkdir -p gpurun_out/r3c
python -m pytest tests/test_ga_wide_gpu.py tests/test_trainer_gpu.py tests/test_ga_gpu.py tests/test_train_gpu.py -x -q -m gpu > gpurun_out/r3c/tests.log 2>&1; echo "tests rc=$?"; tail -30 gpurun_out/r3c/tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3c/bench_drv.json 2> gpurun_out/r3c/bench_drv.err; python -c "
import json; d=json.loads(open('gpurun_out/r3c/bench_drv.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['module_slides_per_s'], d['roofline']['us_per_launch'])"
