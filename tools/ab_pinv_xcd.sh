#!/bin/bash
# A/B of the Moore-Penrose product blocks' placement (TransMIL, N = 100 000): (head, row band) groups dealt over the XCDs (-DTP_HEAD_XCD=0) vs ALL
# blocks of a head on one XCD (=1); interleaved runs of bench.py --workload transmil + the chain kernels' averages.  Run through gpurun
# after tools/build_file_variant.sh headxcd|headgrp transmil_pinv.hip "-DTP_HEAD_XCD=1|0".
out=$GRAFT_REPO_ROOT/gpurun_out/s4/ab_pinv_xcd.txt; mkdir -p $(dirname $out); : > $out
export TMPDIR=/tmp; cd /tmp
for rep in 1 2; do for v in headgrp headxcd; do
  export ACMIL_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libacmil_$v.so
  python $GRAFT_REPO_ROOT/bench.py --workload transmil --steps 60 --warmup 10 --no-cpu-baseline --no-b1 > /tmp/line.json 2>/dev/null
  echo "$v rep $rep: $(python -c "import json;d=json.loads(open('/tmp/line.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])") ms" >> $out
done; done
for v in headgrp headxcd; do
  export ACMIL_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libacmil_$v.so
  rm -rf /tmp/pw; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw -o p -- python $GRAFT_REPO_ROOT/bench.py --workload transmil --steps 30 --warmup 5 --no-cpu-baseline --no-b1 > /dev/null 2>&1
  echo "== $v" >> $out; python - >> $out <<'PY'
import csv,glob
f=glob.glob('/tmp/pw/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'pinv' in r['Name'] or 'attn3x' in r['Name']: print('  %-40s calls %s avg %.2f us' % (r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
cat $out
