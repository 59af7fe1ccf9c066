mkdir -p gpurun_out/r3j
python -m pytest tests/test_transmil_gpu.py tests/test_full_size_gpu.py -q -m gpu -x > gpurun_out/r3j/tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r3j/tests.log
python bench.py --workload transmil --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r3j/tm_fused.json 2>gpurun_out/r3j/tm_fused.err
ACMIL_TM_PINV_CHAIN=1 python bench.py --workload transmil --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r3j/tm_chain.json 2>gpurun_out/r3j/tm_chain.err
for f in tm_fused tm_chain; do python -c "
import json; d=json.loads(open('gpurun_out/r3j/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['executed_frac'])"; done
