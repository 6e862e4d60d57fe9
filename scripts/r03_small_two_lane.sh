#!/bin/bash
# Round 3: 32 / 64 points with two lanes per frame, radix orders by load width and bins per lane (k_tune_pw.hip):
# device parity, then MAG_F32 / DB5 / COMPLEX_F32 rates in the streaming regime.  Output: gpurun_out/r3m/
O=gpurun_out/r3m; mkdir -p $O
export CHECK_MODES=0,1,2,3
for spec in "32 t2a t2b t2c" "64 t2 t2c t2d"; do
  set -- $spec; n=$1; shift
  python scripts/check_variant.py $n "$@" | tail -1 | sed "s/^/N=$n: /" | tee -a $O/check.txt
  vs=$(echo "- $@" | tr ' ' ',')
  for m in 0 2 3; do TUNE_MODE=$m TUNE_SETS=4 TUNE_VARIANTS=$vs python scripts/tune.py $n 2>&1 | tee -a $O/tune.txt; done
done
