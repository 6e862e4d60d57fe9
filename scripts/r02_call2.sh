#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
echo "== tune 8192: v2 ablations"
TUNE_VARIANTS=-,v2,v2s,abl_v2l,abl_v2sl,abl_v2na timeout 300 python scripts/tune.py 8192 2>&1 | tee $OUT/r02_tune_v2_abl.txt
