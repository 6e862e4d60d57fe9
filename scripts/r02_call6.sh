#!/bin/bash
# Round-2 GPU visit: everything after the product / tuning split.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^Written" | tail -12 | tee $OUT/r02_pytest_gpu_a.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== tune, all sizes: product kernels vs round-1 configurations"
TUNE_VARIANTS=-,r1,nd timeout 300 python scripts/tune.py 8192 16384 2>&1 | tee $OUT/r02_tune_sizes.txt
TUNE_VARIANTS=-,r1 timeout 300 python scripts/tune.py 1024 2>&1 | tee -a $OUT/r02_tune_sizes.txt
TUNE_VARIANTS=- timeout 300 python scripts/tune.py 4096 2048 512 256 128 64 32 2>&1 | tee -a $OUT/r02_tune_sizes.txt
echo "== mode rates"
timeout 300 python scripts/mode_rate.py 256 1024 4096 8192 2>&1 | tee $OUT/r02_mode_rates.txt
echo "== bench"
timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 > $OUT/r02_bench_a.json 2> $OUT/r02_bench_a.err
cat $OUT/r02_bench_a.json; tail -3 $OUT/r02_bench_a.err
