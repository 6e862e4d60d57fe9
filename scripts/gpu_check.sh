#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench line, rocprofv3 kernel stats.
# Usage (from the repo root on the GPU box): bash scripts/gpu_check.sh [tag]
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== rocminfo"; /opt/rocm/bin/rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu_$TAG.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke_$TAG.log
echo "== bench"
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
cat $OUT/bench_$TAG.json; tail -3 $OUT/bench_$TAG.err
echo "== rocprofv3 kernel stats"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o bench --output-format csv -- \
  python $R/bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/prof_bench_$TAG.json 2> $OUT/prof_$TAG.err
tail -2 $OUT/prof_$TAG.err
find $OUT/prof_$TAG -name "*kernel_stats*" | head; 
for f in $(find $OUT/prof_$TAG -name "*kernel_stats.csv"); do head -12 $f; cp $f $OUT/kernel_stats_$TAG.csv; done
cat $OUT/prof_bench_$TAG.json
# keep the merged-back payload small
find $OUT/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
