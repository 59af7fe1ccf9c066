ACMIL_WGRAD_TILE=256 python -m pytest tests/test_train_gpu.py tests/test_full_size_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
python -m pytest tests/test_train_gpu.py tests/test_full_size_gpu.py tests/test_ga_wide_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
for t in 128 256; do for n in 10000 24576 50000; do
echo "== tile $t n=$n"; ACMIL_WGRAD_TILE=$t python bench.py --workload train --train-n $n --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done; done
cd /tmp && export TMPDIR=/tmp
for t in 128 256; do for n in 10000 50000; do
  rm -rf /tmp/prof_x; ACMIL_WGRAD_TILE=$t rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -- python $GRAFT_REPO_ROOT/bench.py --workload train --train-n $n --no-cpu-baseline --steps 30 --warmup 5 > /dev/null 2>&1
  echo "tile=$t N=$n"; python $GRAFT_REPO_ROOT/tools/kstats.py /tmp/prof_x | grep -E "wgrad|finish"
done; done
