#!/bin/bash
# Round 4 evidence run, one GPU-box visit: the bench line (driver form, long form), rocprofv3 --kernel-trace --stats of the bench
# command restricted to the headline launches (--no-extra --no-cpu-baseline), the PMC passes of the headline kernel and of its
# windowed twin (separate runs, kernel-trace only), what the window costs, every (size, mode) against the round-2 / round-3
# libraries, the product kernels at every size in the streaming regime, joules per frame.  Output under gpurun_out/r4p/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4p; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
echo "== bench, driver form and long form"
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; head -c 600 $O/bench_driver_form.json; echo
python $R/bench.py --gpus 1 --steps 2000 --warmup 50 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; head -c 400 $O/bench_default.json; echo
echo "== rocprofv3 kernel stats (headline launches only)"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench --output-format csv -- \
  python $R/bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extra > $O/prof_bench.json 2> $O/prof.err
for f in $(find $O/prof -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_bench.csv; head -6 $f; done
cat $O/prof_bench.json | head -c 300; echo
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_w -o bench --output-format csv -- \
  python $R/bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extra --window hann > $O/prof_bench_hann.json 2> $O/prof_w.err
for f in $(find $O/prof_w -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_bench_hann.csv; head -4 $f; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cd $R
echo "== PMC passes: headline kernel"
bash scripts/pmc.sh r04 > $O/pmc.log 2>&1; mkdir -p $O/pmc_rect; cp gpurun_out/pmc_r04/*.summary.txt $O/pmc_rect/ 2>/dev/null; tail -4 $O/pmc.log
echo "== PMC passes: windowed kernel (traffic + instruction counts)"
PMC_EXTRA="--window hann" bash scripts/pmc.sh r04w > $O/pmc_w.log 2>&1; mkdir -p $O/pmc_hann; cp gpurun_out/pmc_r04w/*.summary.txt $O/pmc_hann/ 2>/dev/null; tail -4 $O/pmc_w.log
rm -rf gpurun_out/pmc_r04/*/ gpurun_out/pmc_r04w/*/
echo "== window cost"
timeout 900 python -u scripts/window_rate.py 2>&1 | grep -v amdgpu.ids > $O/window_rate.txt; grep -E " win  " $O/window_rate.txt
echo "== every (size, mode) vs round 2 / round 3"
timeout 900 python -u scripts/ab_modes.py scripts/ab/libfsea_hip_r02.so scripts/ab/libfsea_hip_r03.so 2>&1 | grep -v amdgpu.ids > $O/mode_rates.txt; tail -3 $O/mode_rates.txt
echo "== all sizes, streaming"
TUNE_SETS=4 timeout 900 python scripts/tune.py 8192 1024 4096 16384 2048 512 256 128 64 32 2>&1 | grep -E "variant=-  " | tee $O/tune_all_sizes.txt
echo "== soak 60 s"
timeout 300 python scripts/soak.py 60 random 2>&1 | tail -3 | tee $O/soak.txt
