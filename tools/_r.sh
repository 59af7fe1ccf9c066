export TMPDIR=/tmp
mkdir -p gpurun_out/r04b
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r04b/gpu_tests.log
[ -x build/exp/hbm_calib ] || { mkdir -p build/exp; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o build/exp/hbm_calib tools/hbm_calib.hip; }
python tools/pmc_ga.py --workload transmil --batch 1 --whole-step --steps 10 --out gpurun_out/r04b/pmc > gpurun_out/r04b/pmc_transmil.log 2>&1
cp gpurun_out/r04b/pmc/pmc_*.json gpurun_out/r04b/ 2>/dev/null
for f in gpurun_out/r04b/pmc/pmc_*.json; do cp $f profiles/r04_$(basename $f); done
python bench.py --workload transmil > gpurun_out/r04b/bench_transmil.json 2> gpurun_out/r04b/bench_transmil.log
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tm -o p -- python /root/repo/bench.py --workload transmil --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1)
cp $(find /tmp/prof_tm -name "*kernel_stats.csv" | head -1) gpurun_out/r04b/bench_transmil_kernel_stats.csv
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04b/bench_driver_args.json 2> gpurun_out/r04b/bench_default.log
cat gpurun_out/r04b/gpu_tests.log; tail -c 600 gpurun_out/r04b/bench_transmil.json
