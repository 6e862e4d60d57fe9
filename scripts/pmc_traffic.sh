#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes, kernel-trace only) of the pixel kernels and of
# the fused-stitch sweep.  Usage: bash scripts/pmc_traffic.sh <tag>
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/traffic_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
run() {  # name counter command...
  name=$1; ctr=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/$name -o pmc --output-format csv -- "$@" > $OUT/$name.out 2> $OUT/$name.err
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/scripts/pmc_summary.py $f | sed "s/^/$name: /"; rm -f $f; else echo "no counter file for $name"; tail -3 $OUT/$name.err; fi
  find $OUT/$name -name "*kernel_trace.csv" -delete
}
for c in FETCH_SIZE WRITE_SIZE; do
  run broad_$c $c python $R/bench.py --workload broad --steps 5 --warmup 2
  run modes_$c $c python $R/scripts/mode_rate.py 256 4096
done
