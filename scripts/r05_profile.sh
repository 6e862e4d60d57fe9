#!/bin/bash
# Round 5 evidence run, one GPU-box visit: the bench line in the driver's form, rocprofv3 --kernel-trace --stats of the headline
# launches, the PMC passes of the headline kernel (separate runs, kernel-trace only) from which profiles/traffic.json and the
# line's valu_issue_frac / lds_active_frac are made, and the driver's command lines at 1, 2, 4, 8 ranks on the one GPU (gloo).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5p; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
echo "== rocprofv3 kernel stats (headline launches only)"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench --output-format csv -- \
  python $R/bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extra > $O/prof_bench.json 2> $O/prof.err
for f in $(find $O/prof -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_bench.csv; head -6 $f; done
head -c 300 $O/prof_bench.json; echo
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cd $R
echo "== PMC passes: headline kernel"
bash scripts/pmc.sh r05 > $O/pmc.log 2>&1
cat gpurun_out/pmc_r05/*.summary.txt > $O/pmc_fsea_fft8192_u8_mag.txt 2>/dev/null; tail -4 $O/pmc.log
rm -rf gpurun_out/pmc_r05/*/
echo "== the driver's command lines, 1 / 2 / 4 / 8 ranks on this GPU"
bash scripts/r05_ranks_check.sh > $O/ranks.log 2>&1; grep -E "^==|multi_gpu_error|checksum =|wall|value" gpurun_out/r05_ranks_on_one_gpu.txt | cut -c1-220
