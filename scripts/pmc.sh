#!/bin/bash
# PMC counter passes (each in its own rocprofv3 run, kernel-trace only) over a short bench.
# Usage: bash scripts/pmc.sh <tag> [workload]
TAG=${1:-r01}
WL=${2:-batch8192x4096}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
run() {  # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o pmc --output-format csv -- \
    python $R/bench.py --gpus 1 --steps 20 --warmup 4 --no-cpu-baseline --no-extra --workload $WL $PMC_EXTRA > $OUT/$name.json 2> $OUT/$name.err
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/scripts/pmc_summary.py $f > $OUT/$name.summary.txt; cat $OUT/$name.summary.txt; rm -f $f; else echo "no counter file for $name"; tail -3 $OUT/$name.err; fi
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
