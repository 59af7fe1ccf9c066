timeout 900 python -m pytest tests/test_transmil_gpu.py -x -q -m gpu 2>&1 | tail -3
b() { tag=$1; shift; for i in 1 2; do env "$@" python bench.py --workload transmil --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])"; done; }
b default X=1
b serial ACMIL_TM_SIDE_STREAM=0
