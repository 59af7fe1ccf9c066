#!/bin/bash
# TransMIL bench line + rocprofv3 kernel stats of the same command into gpurun_out/$1/ (run on the GPU box):  tools/tm_stats.sh r04c [env assignments...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
env "$@" python bench.py --workload transmil --no-cpu-baseline > $OUT/tm.json 2> $OUT/tm.err
python -c "
import json; d=json.loads(open('$OUT/tm.json').read().strip().splitlines()[-1]); print('transmil ms_per_step', d['ms_per_step'])"
export TMPDIR=/tmp; rm -rf /tmp/p_$TAG
(cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$TAG -o p -- python $ROOT/bench.py --workload transmil --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1)
cp $(find /tmp/p_$TAG -name "*kernel_stats.csv" | head -1) $OUT/tm_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/tm_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:24]:
    print("%-60s calls %5s avg %8.1f us  per-fwd %8.1f us %5.1f%%" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3/35, 100*float(r['TotalDurationNs'])/tot))
print("sum per forward (us):", round(tot/1e3/35,1))
PY
