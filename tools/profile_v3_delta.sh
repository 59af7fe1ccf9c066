#!/bin/bash
# After a change to ga_forward_kernel_v3.h only: the evidence of the families it serves (UNI, CLIP-L) + the grouped GigaPath PMC, into
# gpurun_out/<tag>/ beside a full tools/profile_round.sh run of the same tag.  Run through gpurun.
TAG=${1:-rXX}
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
[ -x build/exp/hbm_calib ] || { mkdir -p build/exp; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o build/exp/hbm_calib tools/hbm_calib.hip; }
for w in ga_uni ga_clip_l; do python tools/pmc_ga.py --workload $w --batch 64 --steps 8 --out $OUT/pmc > $OUT/pmc_$w.log 2>&1; done
python tools/pmc_ga.py --workload ga_gigapath --batch 16 --whole-step --steps 8 --out $OUT/pmc > $OUT/pmc_ga_gigapath_b16.log 2>&1
cp $OUT/pmc/pmc_*.json $OUT/ 2>/dev/null
for f in $OUT/pmc/pmc_*.json; do cp $f profiles/${TAG}_$(basename $f); done
for w in ga_uni ga_clip_l; do python bench.py --workload $w --steps 20 --warmup 5 > $OUT/bench_$w.json 2> $OUT/bench_$w.log; done
python bench.py --workload ga_gigapath --steps 30 --warmup 5 > $OUT/bench_ga_gigapath.json 2> $OUT/bench_ga_gigapath.log
for w in ga_uni ga_clip_l; do
  d=/tmp/prof_$w; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $ROOT/bench.py --workload $w --steps 10 --warmup 3 --no-b1 --no-cpu-baseline > /dev/null 2> $OUT/${w}_rocprof.log)
  f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/bench_${w}_f16x3_b64_kernel_stats.csv
done
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_default.log
tail -c 600 $OUT/bench_ga_uni.json; echo
