#!/bin/bash
# A/B of the backward tile height on a group step (8 x 50 000 rows): 64-row tiles (default) vs 32-row tiles whose gate pass keeps the staged
# fp32 h rows in registers (HKEEP) -- kernel averages under rocprofv3 --kernel-trace --stats; run through gpurun
out=$GRAFT_REPO_ROOT/gpurun_out/s4/ab_bwd_rows.txt; mkdir -p $(dirname $out); : > $out
export ACMIL_HIP_LIB=$GRAFT_REPO_ROOT/acmil_amd/libacmil_hip_ab.so TMPDIR=/tmp
cd /tmp
for v in "" "ACMIL_GA_BWD_ROWS=32"; do
  rm -rf /tmp/pw
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw -o p -- python $GRAFT_REPO_ROOT/bench.py --workload train --train-n 50000 --bags-per-step 8 --steps 30 --warmup 8 --no-cpu-baseline > /tmp/line.json 2>/dev/null
  echo "== [$v] $(python -c "import json;d=json.loads(open('/tmp/line.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'])")" >> $out
  grep -E "bwd_tile|ga_opt_step|wgrad" $(find /tmp/pw -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4 >> $out
done
cat $out
