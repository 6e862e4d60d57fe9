#!/bin/bash
# Round 5: rocprofv3 evidence for BASELINE config 4 on one GPU (the fused-stitch sweep, fsea_fft4096_u8_db5 through
# fsea_exec_u8_tiled_device) on the final build: kernel stats of `bench.py --workload broad`, then FETCH_SIZE / WRITE_SIZE
# and the SQ instruction / LDS counters in separate PMC passes (kernel-trace only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5b; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o broad --output-format csv -- \
  python $R/bench.py --workload broad --steps 200 --warmup 10 --no-extra > $O/bench_broad.json 2> $O/prof.err
for f in $(find $O/prof -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_broad.csv; head -4 $f; done
head -c 400 $O/bench_broad.json; echo
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
run() {  # name counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o pmc --output-format csv -- \
    python $R/bench.py --workload broad --steps 20 --warmup 3 --no-extra > $O/$name.json 2> $O/$name.err
  f=$(find $O/$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/scripts/pmc_summary.py $f | grep db5; rm -f $f; else echo "no counter file for $name"; tail -3 $O/$name.err; fi
  find $O/$name -name "*kernel_trace.csv" -delete
}
{
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVES
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU
} | tee $O/pmc_fsea_fft4096_u8_db5_sweep.txt
