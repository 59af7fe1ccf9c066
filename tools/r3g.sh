mkdir -p gpurun_out/r3g
V=$PWD/build/variants
GA_PROF=1 python tools/probe_ga.py --iters 5 --rounds 3 --out gpurun_out/r3g/probe --variants "prof:ACMIL_HIP_LIB=$V/libacmil_prof.so;prof_pair:ACMIL_HIP_LIB=$V/libacmil_prof.so,ACMIL_GA2_PAIR=1;prof_nodma:ACMIL_HIP_LIB=$V/libacmil_prof_nodma.so;prof_nodma_pair:ACMIL_HIP_LIB=$V/libacmil_prof_nodma.so,ACMIL_GA2_PAIR=1;prof_nomfma:ACMIL_HIP_LIB=$V/libacmil_prof_nomfma.so;prof_w8:ACMIL_HIP_LIB=$V/libacmil_prof.so,ACMIL_GA2_WAVES=8" 2>&1 | grep "^PROF" > gpurun_out/r3g/prof.log
cat gpurun_out/r3g/prof.log
: > gpurun_out/r3g/clk.log
for v in abl15 abl12 abl3 abl13 abl14; do
  ACMIL_HIP_LIB=$V/libacmil_$v.so python tools/abl_clock.py $v 16 >> gpurun_out/r3g/clk.log 2>&1
  ACMIL_GA2_PAIR=1 ACMIL_HIP_LIB=$V/libacmil_$v.so python tools/abl_clock.py ${v}_pair 16 >> gpurun_out/r3g/clk.log 2>&1
done
ACMIL_GA2_WAVES=8 ACMIL_HIP_LIB=$V/libacmil_abl3.so python tools/abl_clock.py abl3_w8 16 >> gpurun_out/r3g/clk.log 2>&1
ACMIL_GA2_WAVES=8 ACMIL_HIP_LIB=$V/libacmil_abl12.so python tools/abl_clock.py abl12_w8 16 >> gpurun_out/r3g/clk.log 2>&1
grep ABLCLK gpurun_out/r3g/clk.log | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l.split('ABLCLK ')[1]); print('%-14s %7.1f us  %4d MHz  %4d W  %.3f Mcyc' % (d['name'], d['us_per_launch'], d['sclk_mhz_mean'], d['power_w_mean'], d['us_per_launch']*d['sclk_mhz_mean']/1e6))"
