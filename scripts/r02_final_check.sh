#!/bin/bash
# Round-2 evidence run on the final product kernels: tests, bench line, rocprofv3 stats, PMC, tune tables.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^Written\|^Composing\|^Frequency" | tail -8 | tee $OUT/${TAG}_pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench (as the driver runs it, then a longer one)"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_form.json 2> $OUT/${TAG}_bench.err; cat $OUT/${TAG}_bench_driver_form.json | cut -c1-1400
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 > $OUT/${TAG}_bench_line.json 2>> $OUT/${TAG}_bench.err; cat $OUT/${TAG}_bench_line.json | cut -c1-900
echo "== rocprofv3 kernel stats of the bench command"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o bench --output-format csv -- \
  python $R/bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
for f in $(find $OUT/${TAG}_prof -name "*kernel_stats.csv"); do head -12 $f; cp $f $OUT/${TAG}_kernel_stats_bench.csv; done
cut -c1-700 $OUT/${TAG}_prof_bench.json
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -size +20M -delete
cd $R
echo "== PMC passes"
bash scripts/pmc.sh ${TAG} > $OUT/${TAG}_pmc.log 2>&1; grep "fsea_fft8192_u8_mag" $OUT/${TAG}_pmc.log | head -40
echo "== tune: product kernels vs round-1 configurations, steady state (one 768 MiB set) and streaming (4 sets)"
TUNE_VARIANTS=-,cp0,r1 timeout 300 python scripts/tune.py 8192 16384 1024 2>&1 | grep variant | tee $OUT/${TAG}_tune_sizes.txt
TUNE_VARIANTS=- timeout 300 python scripts/tune.py 4096 2048 512 256 128 64 32 2>&1 | grep variant | tee -a $OUT/${TAG}_tune_sizes.txt
echo "-- streaming" | tee -a $OUT/${TAG}_tune_sizes.txt
TUNE_SETS=4 TUNE_VARIANTS=-,cp0,r1 timeout 300 python scripts/tune.py 8192 16384 1024 2>&1 | grep variant | tee -a $OUT/${TAG}_tune_sizes.txt
TUNE_SETS=4 TUNE_VARIANTS=-,cp0 timeout 300 python scripts/tune.py 4096 2048 256 2>&1 | grep variant | tee -a $OUT/${TAG}_tune_sizes.txt
TUNE_SETS=4 TUNE_VARIANTS=- timeout 300 python scripts/tune.py 512 128 64 32 2>&1 | grep variant | tee -a $OUT/${TAG}_tune_sizes.txt
echo "== mode rates"
timeout 300 python scripts/mode_rate.py 256 1024 4096 8192 2>&1 | tee $OUT/${TAG}_mode_rates.txt
echo "== workgroup trace"
timeout 120 python scripts/wg_trace.py 8192 4096 2>&1 | grep -v Warning | tee $OUT/${TAG}_wg_trace.txt
echo "== energy"
ENERGY_SECONDS=3 timeout 600 python scripts/energy_probe.py - cp0 r1 abl_io 2>&1 | tee $OUT/${TAG}_energy_final.txt
