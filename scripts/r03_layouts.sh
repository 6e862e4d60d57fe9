#!/bin/bash
# Round 3: radix orders with wider pass-0 loads / more adjacent bins per lane at every size (tuning variants of k_tune_lay.hip):
# device parity, then MAG_F32 and DB5 rates in the streaming regime.  Output: gpurun_out/r3l/
O=gpurun_out/r3l; mkdir -p $O
export CHECK_MODES=0,2,3
for spec in "512 888 1632" "256 488 884" "128 448" "2048 81616" "4096 16328 83216" "8192 163216 83232"; do
  set -- $spec; n=$1; shift
  python scripts/check_variant.py $n "$@" | tail -1 | sed "s/^/N=$n: /" | tee -a $O/check.txt
  vs=$(echo "- $@" | tr ' ' ',')
  for m in 0 2; do TUNE_MODE=$m TUNE_SETS=4 TUNE_VARIANTS=$vs python scripts/tune.py $n 2>&1 | tee -a $O/tune.txt; done
done
