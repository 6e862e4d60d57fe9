#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
V=-,abl_v2l,v2s,abl_nolds,abl_noflop,abl_io,abl_nostore,abl_noload,abl_nomag
echo "== noise-like input"
TUNE_VARIANTS=$V timeout 300 python scripts/tune.py 8192 2>&1 | tee $OUT/r02_tune_abl_noise.txt
echo "== constant input (no toggling: not power-limited)"
TUNE_CONST_INPUT=1 TUNE_VARIANTS=$V timeout 300 python scripts/tune.py 8192 2>&1 | tee $OUT/r02_tune_abl_const.txt
echo "== clocks (wg_trace, 32768-frame launches)"
for v in "" abl_v2l abl_nolds abl_noflop abl_io; do
  echo "-- variant '$v'"; FSEA_VARIANT=$v timeout 120 python scripts/wg_trace.py 8192 32768 2>&1 | grep -E "event time|shader clock|mean time of iteration"
done
