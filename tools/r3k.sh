mkdir -p gpurun_out/r3k
python -m pytest tests/test_transmil_gpu.py tests/test_full_size_gpu.py -q -m gpu -x > gpurun_out/r3k/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3k/tests.log
python bench.py --workload transmil --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r3k/tm_aux.json 2>gpurun_out/r3k/tm_aux.err
ACMIL_TM_NO_AUX=1 python bench.py --workload transmil --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r3k/tm_noaux.json 2>gpurun_out/r3k/tm_noaux.err
ACMIL_TM_PINV_FUSED=1 python bench.py --workload transmil --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r3k/tm_aux_fused.json 2>gpurun_out/r3k/tm_aux_fused.err
for f in tm_aux tm_noaux tm_aux_fused; do python -c "
import json; d=json.loads(open('gpurun_out/r3k/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['executed_frac'])"; done
