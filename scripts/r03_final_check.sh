#!/bin/bash
# Round 3, one GPU-box visit at the end of the round: the GPU test tier, smoke, the bench line in the driver's form and in
# the long form (with the two-stream figure), rocprofv3 kernel stats of the bench command, the PMC passes of the headline
# kernel, per-mode rates and the all-sizes table.  Output under gpurun_out/r3f/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3f; mkdir -p $O
cd $R; export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.log
cd /tmp
echo "== bench (driver form)"
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; cat $O/bench_driver_form.json
echo "== bench (long form, two streams beside)"
python $R/bench.py --gpus 1 --steps 2000 --warmup 50 --no-cpu-baseline --two-stream > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json
echo "== rocprofv3 kernel stats"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench --output-format csv -- \
  python $R/bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err
for f in $(find $O/prof -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_bench.csv; head -12 $f; done
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.db" -delete
cd $R
echo "== PMC (headline kernel)"
bash scripts/pmc.sh r03f > $O/pmc.log 2>&1; cp -r gpurun_out/pmc_r03f/*.summary.txt $O/ 2>/dev/null; tail -5 $O/pmc.log
echo "== mode rates, all sizes"
python scripts/mode_rate.py 256 1024 4096 8192 > $O/mode_rates.txt 2>&1; tail -30 $O/mode_rates.txt
TUNE_SETS=4 python scripts/tune.py 8192 1024 4096 16384 2048 512 256 128 64 32 2>&1 | grep -E "variant=-  " | tee $O/tune_all_sizes.txt
python scripts/nrf_latency.py 2>&1 | tee $O/nrf_latency.txt
rm -rf gpurun_out/pmc_r03f/*/
