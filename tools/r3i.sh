mkdir -p gpurun_out/r3i
python -m pytest tests -q -m gpu -x > gpurun_out/r3i/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3i/tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3i/bench_drv.json 2> gpurun_out/r3i/bench_drv.err; python -c "
import json; d=json.loads(open('gpurun_out/r3i/bench_drv.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['module_slides_per_s'], d['roofline']['us_per_launch'], d['roofline']['executed_frac'], d['attention_fwd_ms_per_slide_b1'])"
python bench.py --gpus 1 --steps 20 --warmup 5 --batch 16 --no-cpu-baseline > gpurun_out/r3i/bench_b16.json 2> gpurun_out/r3i/bench_b16.err; python -c "
import json; d=json.loads(open('gpurun_out/r3i/bench_b16.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['module_slides_per_s'], d['roofline']['us_per_launch'])"
