#!/bin/bash
# Round 3: the last butterfly level in power form (FftCfg OPT 8388608; tuning variants "pw" of k_tune_pw.hip, "f8" = the
# small sizes' product + fused last-pass twiddles): device parity, then MAG_F32 / DB5 / DB10 rates in the streaming regime.
# Output: gpurun_out/r3w/
O=gpurun_out/r3w; mkdir -p $O
export CHECK_MODES=0,1,2
for spec in "8192 pw" "4096 pw" "1024 pw" "256 pw f8" "16384 pw" "2048 pw" "512 pw f8" "128 pw"; do
  set -- $spec; n=$1; shift
  python scripts/check_variant.py $n "$@" | tail -1 | sed "s/^/N=$n: /" | tee -a $O/check.txt
  vs=$(echo "- $@" | tr ' ' ',')
  for m in 0 2 1; do TUNE_MODE=$m TUNE_SETS=4 TUNE_VARIANTS=$vs python scripts/tune.py $n 2>&1 | tee -a $O/tune.txt; done
done
