mkdir -p gpurun_out/r3l
python -m pytest tests/test_ga_wide_gpu.py tests/test_variants_gpu.py tests/test_linear_gpu.py -q -m gpu -x > gpurun_out/r3l/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3l/tests.log
for w in ga_uni ga_gigapath ga_clip_l; do python bench.py --workload $w --steps 30 --warmup 5 > gpurun_out/r3l/$w.json 2>gpurun_out/r3l/$w.err; python -c "
import json; d=json.loads(open('gpurun_out/r3l/$w.json').read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d['roofline']['executed_frac'], d['roofline']['projection_kernel'], d.get('max_abs_err_vs_oracle'), d.get('cpu_baseline',{}).get('value'))"; done
