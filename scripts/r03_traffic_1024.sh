#!/bin/bash
# HBM traffic of the kernels whose store pattern changed late in round 3: 1024 points (8 x 16 x 8: 16-byte f32 row stores, dword
# pixel stores), COMPLEX_F32 rows at 1024 / 2048 (default cache policy for the split 16-byte stores), and the half-overlap kernel
# of config 5.  FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes (kernel-trace only).  Output: gpurun_out/r3t/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3t
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
run() {  # name counter command...
  name=$1; ctr=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/$name -o pmc --output-format csv -- "$@" > $OUT/$name.out 2> $OUT/$name.err
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/scripts/pmc_summary.py $f | sed "s/^/$name: /"; rm -f $f; else echo "no counter file for $name"; tail -3 $OUT/$name.err; fi
  find $OUT/$name -name "*kernel_trace.csv" -delete; find $OUT/$name -name "*.db" -delete
}
for c in FETCH_SIZE WRITE_SIZE; do
  run modes_$c $c python $R/scripts/mode_rate.py 1024 2048
  run stft_$c $c python $R/bench.py --workload stft16384x8191 --steps 5 --warmup 2 --no-cpu-baseline --no-extra
done
