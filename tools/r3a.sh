mkdir -p gpurun_out/r3a
( python tools/abl_clock.py base 16 ) > gpurun_out/r3a/abl.log 2>&1
for v in abl1 abl2 abl3 abl4 abl8 abl12 abl15; do
  ACMIL_HIP_LIB=$PWD/build/variants/libacmil_$v.so timeout 120 python tools/abl_clock.py $v 16 >> gpurun_out/r3a/abl.log 2>&1
done
python tools/abl_clock.py base_b1 1 >> gpurun_out/r3a/abl.log 2>&1
python tools/abl_clock.py base_f16 16 float16 >> gpurun_out/r3a/abl.log 2>&1
grep ABLCLK gpurun_out/r3a/abl.log
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3a/bench_drv$i.json 2>gpurun_out/r3a/bench_drv$i.err; done
python bench.py --no-cpu-baseline > gpurun_out/r3a/bench_100.json 2>&1
for f in gpurun_out/r3a/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['attention_fwd_ms_per_slide_b1'])"; done
