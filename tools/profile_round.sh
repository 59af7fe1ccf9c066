#!/bin/bash
# Collect the round's bench lines + rocprofv3 evidence on the GPU box (run through gpurun):  tools/profile_round.sh r02
# Writes gpurun_out/<tag>/: bench JSON lines, kernel-trace stats CSVs, PMC summaries (tools/pmc_ga.py).  Copy what is to be
# judged into profiles/.
TAG=${1:-rXX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
run_stats() {   # name, bench args...
  name=$1; shift
  d=/tmp/prof_$name; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $ROOT/bench.py "$@" > $OUT/${name}_profiled.json 2> $OUT/${name}_rocprof.log)
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv
}
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log
python bench.py --workload ga_cfg3 > $OUT/bench_ga_cfg3.json 2> $OUT/bench_ga_cfg3.log
python bench.py --batch 1 --no-cpu-baseline > $OUT/bench_b1.json 2>> $OUT/bench_default.log
run_stats bench_f16x3_b16 --steps 50 --warmup 5 --no-b1 --no-cpu-baseline
run_stats bench_ga_cfg3 --workload ga_cfg3 --steps 50 --warmup 5 --no-b1 --no-cpu-baseline
run_stats bench_transmil --workload transmil --steps 30 --warmup 5 --no-cpu-baseline
run_stats bench_train_n10k --workload train --steps 200 --warmup 20 --no-cpu-baseline
run_stats bench_train_n50k --workload train --train-n 50000 --steps 200 --warmup 20 --no-cpu-baseline
python bench.py --workload transmil > $OUT/bench_transmil.json 2> $OUT/bench_transmil.log
python bench.py --workload train > $OUT/bench_train_n10k.json 2> $OUT/bench_train.log
python bench.py --workload train --train-n 50000 > $OUT/bench_train_n50k.json 2>> $OUT/bench_train.log
python tools/pmc_ga.py --batch 16 --out $OUT/pmc > $OUT/pmc_ga_eval.log 2>&1
python tools/pmc_ga.py --workload ga_cfg3 --batch 16 --out $OUT/pmc > $OUT/pmc_ga_cfg3.log 2>&1
cp $OUT/pmc/pmc_*.json $OUT/ 2>/dev/null
tail -c 1500 $OUT/bench_default.json; echo; tail -c 1200 $OUT/bench_ga_cfg3.json; echo
head -5 $OUT/bench_f16x3_b16_kernel_stats.csv
ls $OUT
