#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
ENERGY_SECONDS=3 timeout 600 python scripts/energy_probe.py - st_nt st_sc1 st_sc0sc1 st_sc1nt 2>&1 | tee $OUT/r02_energy_store_policy.txt
for v in "" st_nt st_sc1 st_sc1nt; do
  FSEA_BENCH_VARIANT=$v timeout 300 python bench.py --gpus 1 --steps 400 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant=%-9s value %.2f Mframes/s launch %.5f ms frac %.4f' % ('${v:--}', d['value']/1e6, d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done
