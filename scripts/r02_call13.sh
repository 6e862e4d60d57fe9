#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
echo "== streaming regime (4 rotating buffer sets of 768 MiB), MAG"
TUNE_SETS=4 TUNE_VARIANTS=-,st_nt,ldst_nt timeout 600 python scripts/tune.py 8192 16384 4096 2048 1024 256 2>&1 | tee $OUT/r02_tune_nt_streaming.txt
echo "== streaming regime, DB5 pixels (run-time-mode kernel for the variants)"
TUNE_MODE=2 TUNE_SETS=4 TUNE_VARIANTS=-,st_nt,ldst_nt timeout 600 python scripts/tune.py 4096 1024 256 2>&1 | tee -a $OUT/r02_tune_nt_streaming.txt
echo "== BASELINE launch size: 4096 frames of 8192 (6 sets), 32768 of 1024"
TUNE_FRAMES=4096 TUNE_SETS=6 TUNE_VARIANTS=-,st_nt,ldst_nt timeout 600 python scripts/tune.py 8192 2>&1 | tee -a $OUT/r02_tune_nt_streaming.txt
TUNE_FRAMES=32768 TUNE_SETS=6 TUNE_VARIANTS=-,st_nt,ldst_nt timeout 600 python scripts/tune.py 1024 2>&1 | tee -a $OUT/r02_tune_nt_streaming.txt
TUNE_FRAMES=8191 TUNE_SETS=3 TUNE_VARIANTS=-,st_nt,ldst_nt timeout 600 python scripts/tune.py 16384 2>&1 | tee -a $OUT/r02_tune_nt_streaming.txt
