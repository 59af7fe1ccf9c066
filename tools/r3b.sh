mkdir -p gpurun_out/r3b
python -m pytest tests/test_ga_gpu.py tests/test_trainer_gpu.py tests/test_full_size_gpu.py -x -q -m gpu > gpurun_out/r3b/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r3b/tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3b/bench_drv.json 2> gpurun_out/r3b/bench_drv.err; tail -c 3000 gpurun_out/r3b/bench_drv.json
