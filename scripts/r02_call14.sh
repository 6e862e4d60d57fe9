#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
echo "== streaming regime, DB5 pixels: product kernel (*_u8_db5) vs run-time-mode variants with nt policies"
TUNE_MODE=2 TUNE_SETS=4 TUNE_VARIANTS=-,r1,st_nt,ldst_nt timeout 600 python scripts/tune.py 8192 2>&1 | grep variant | tee $OUT/r02_tune_nt_modes.txt
TUNE_MODE=2 TUNE_SETS=4 TUNE_VARIANTS=-,x0,st_nt,ldst_nt timeout 600 python scripts/tune.py 4096 1024 2>&1 | grep variant | tee -a $OUT/r02_tune_nt_modes.txt
TUNE_MODE=2 TUNE_SETS=4 TUNE_VARIANTS=-,st_nt,ldst_nt timeout 600 python scripts/tune.py 256 2>&1 | grep variant | tee -a $OUT/r02_tune_nt_modes.txt
echo "== complex output"
TUNE_MODE=3 TUNE_SETS=3 TUNE_VARIANTS=-,st_nt,ldst_nt timeout 600 python scripts/tune.py 8192 1024 2>&1 | grep variant | tee -a $OUT/r02_tune_nt_modes.txt
