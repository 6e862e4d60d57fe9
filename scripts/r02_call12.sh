#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
ENERGY_SECONDS=3 timeout 600 python scripts/energy_probe.py - st_nt ld_nt ldst_nt 2>&1 | tee $OUT/r02_energy_store_policy2.txt
TUNE_VARIANTS=-,st_nt,ld_nt,ldst_nt timeout 300 python scripts/tune.py 8192 2>&1 | tee $OUT/r02_tune_nt.txt
TUNE_VARIANTS=-,st_nt timeout 300 python scripts/tune.py 16384 4096 2048 1024 256 2>&1 | tee -a $OUT/r02_tune_nt.txt
for v in "" st_nt ldst_nt; do
  FSEA_BENCH_VARIANT=$v timeout 300 python bench.py --gpus 1 --steps 400 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant=%-9s value %.2f Mframes/s launch %.5f ms frac %.4f' % ('${v:--}', d['value']/1e6, d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done
