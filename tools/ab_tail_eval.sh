#!/bin/bash
# A/B: eval finish as merge + heads (two launches) vs the single tail launch (ACMIL_GA_TAIL_EVAL=1, A/B build) -- run through gpurun
set -u
out=gpurun_out/s4/ab_tail.txt; : > $out
export ACMIL_HIP_LIB=$PWD/acmil_amd/libacmil_hip_ab.so
for rep in 1 2; do
  echo "== two launches (rep $rep)" >> $out; python tools/time_single_bag.py >> $out 2>&1
  echo "== tail eval (rep $rep)" >> $out; ACMIL_GA_TAIL_EVAL=1 python tools/time_single_bag.py >> $out 2>&1
done
cat $out
