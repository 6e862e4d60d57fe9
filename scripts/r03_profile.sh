#!/bin/bash
# Round 3 evidence run: rocprofv3 --kernel-trace --stats of the bench command (driver form), the PMC passes of the headline
# kernel (separate runs, kernel-trace only), HBM traffic of the pixel kernels and the sweep.  Output under gpurun_out/r3p/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3p; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
python $R/bench.py --gpus 1 --steps 2000 --warmup 50 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench --output-format csv -- \
  python $R/bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err
for f in $(find $O/prof -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_bench.csv; head -14 $f; done
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.db" -delete
cd $R
bash scripts/pmc.sh r03 > $O/pmc.log 2>&1; cp -r gpurun_out/pmc_r03/*.summary.txt $O/ 2>/dev/null
bash scripts/pmc_traffic.sh r03 > $O/pmc_traffic.log 2>&1; cat $O/pmc_traffic.log | tail -30
python scripts/mode_rate.py 256 1024 4096 8192 > $O/mode_rates.txt 2>&1
TUNE_SETS=4 python scripts/tune.py 8192 1024 4096 16384 2048 512 256 128 64 32 2>&1 | grep -E "variant=-  " > $O/tune_all_sizes.txt
rm -rf gpurun_out/pmc_r03/*/ gpurun_out/traffic_r03/*/
