#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
TUNE_VARIANTS=-,pr,df,dfpr timeout 300 python scripts/tune.py 8192 2>&1 | tee $OUT/r02_tune_prio.txt
for v in "" pr dfpr; do
  FSEA_BENCH_VARIANT=$v timeout 300 python bench.py --gpus 1 --steps 400 --warmup 20 --no-extra --no-cpu-baseline > $OUT/r02_bench_p_${v:-def}.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$OUT/r02_bench_p_${v:-def}.json").read().strip().splitlines()[-1])
print("variant=%-5s value %.2f Mframes/s  ms/step %.5f  launch %.5f ms  frac %.4f" % ("${v:-def}", d["value"]/1e6, d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
PY
done
for v in "" pr; do FSEA_VARIANT=$v timeout 120 python scripts/wg_trace.py 8192 4096 2>&1 | grep -E "event time|end  :|workgroup duration|idle tail|mean time of iter|shader clock"; done
