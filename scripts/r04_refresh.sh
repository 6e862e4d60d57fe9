#!/bin/bash
# Refresh of the windowed kernel's evidence after its tables moved in front of the first input bytes; a 4-minute soak.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4r; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_w -o bench --output-format csv -- \
  python $R/bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extra --window hann > $O/prof_bench_hann.json 2> $O/prof_w.err
for f in $(find $O/prof_w -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_bench_hann.csv; head -3 $f; done
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench --output-format csv -- \
  python $R/bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extra > $O/prof_bench.json 2> $O/prof.err
for f in $(find $O/prof -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_bench.csv; head -3 $f; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cd $R
PMC_EXTRA="--window hann" bash scripts/pmc.sh r04w > $O/pmc_w.log 2>&1; mkdir -p $O/pmc_hann; cp gpurun_out/pmc_r04w/*.summary.txt $O/pmc_hann/ 2>/dev/null; rm -rf gpurun_out/pmc_r04w/*/
cat $O/pmc_hann/fetch.summary.txt $O/pmc_hann/write.summary.txt
timeout 400 python scripts/soak.py 240 random 2>&1 | tail -2 | tee $O/soak.txt
