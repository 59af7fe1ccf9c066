mkdir -p gpurun_out/r3m
python -m pytest tests -q -m gpu -x > gpurun_out/r3m/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3m/tests.log
python bench.py --workload train --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r3m/train10k.json 2>gpurun_out/r3m/train10k.err
python bench.py --workload train --train-n 50000 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r3m/train50k.json 2>gpurun_out/r3m/train50k.err
for f in train10k train50k; do python -c "
import json; d=json.loads(open('gpurun_out/r3m/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
