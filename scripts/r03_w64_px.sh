#!/bin/bash
# Round 3, item 1: the single-wave 4096-point layouts and the v_cvt_pk_u8_f32 pixel epilogue, parity + rates.
# Writes gpurun_out/r3b/*.txt
set -u
O=gpurun_out/r3b; mkdir -p $O
export CHECK_MODES=0,1,2
python scripts/check_variant.py 4096 w64 w64b pk px0 > $O/check_4096.txt 2>&1; tail -1 $O/check_4096.txt
python scripts/check_variant.py 8192 pk px0 > $O/check_8192.txt 2>&1; tail -1 $O/check_8192.txt
python scripts/check_variant.py 256 pk px0 p64 > $O/check_256.txt 2>&1; tail -1 $O/check_256.txt
python scripts/check_variant.py 1024 px0 > $O/check_1024.txt 2>&1; tail -1 $O/check_1024.txt
for m in 2 1; do
  TUNE_MODE=$m TUNE_SETS=4 TUNE_VARIANTS=-,px0,pk,w64,w64b,s2 python scripts/tune.py 4096 2>&1 | tee -a $O/tune_px.txt
  TUNE_MODE=$m TUNE_SETS=4 TUNE_VARIANTS=-,px0,pk python scripts/tune.py 8192 2>&1 | tee -a $O/tune_px.txt
  TUNE_MODE=$m TUNE_SETS=4 TUNE_VARIANTS=-,px0,pk,p64,p16 python scripts/tune.py 256 2>&1 | tee -a $O/tune_px.txt
  TUNE_MODE=$m TUNE_SETS=4 TUNE_VARIANTS=-,px0 python scripts/tune.py 1024 2>&1 | tee -a $O/tune_px.txt
done
TUNE_MODE=0 TUNE_SETS=4 TUNE_VARIANTS=-,w64,s2 python scripts/tune.py 4096 2>&1 | tee -a $O/tune_mag.txt
TUNE_MODE=0 TUNE_SETS=4 TUNE_VARIANTS=-,p64,p16 python scripts/tune.py 256 2>&1 | tee -a $O/tune_mag.txt
