#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
echo "== microbenchmarks"
scripts/ubench/bin/cvt_pk_u8 2>&1 | tee $OUT/r02_cvt_pk_u8.txt | tail -12
scripts/ubench/bin/exchange_lds_vs_dpp 2>&1 | tee $OUT/r02_exchange_dpp_vs_lds.txt
echo "== PMC passes (traffic, SQ counters)"
bash scripts/pmc.sh r02a > $OUT/r02_pmc_a.log 2>&1; tail -40 $OUT/r02_pmc_a.log
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^Written" | tail -25 | tee $OUT/r02_pytest_gpu_b.log
