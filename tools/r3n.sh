mkdir -p gpurun_out/r3n; export TMPDIR=/tmp
d=/tmp/prof_tm; rm -rf $d
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/bench.py --workload transmil --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1)
f=$(find $d -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r3n/tm_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3n/tm_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print('%-60s %6s %9.1f us %5.1f%%' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/tot*100))
PY
ACMIL_TM_GENERIC_GEMM=1 python bench.py --workload transmil --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('generic gemm', d['value'], d['ms_per_step'])"
python bench.py --workload transmil --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('packed lin', d['value'], d['ms_per_step'])"
