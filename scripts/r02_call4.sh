#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
echo "== 8192: packed-add +-i (default now) vs fma form, deferred twiddles, late ticket"
TUNE_VARIANTS=-,fmai,df,tk,dftk timeout 300 python scripts/tune.py 8192 2>&1 | tee $OUT/r02_tune_k124.txt
echo "== other sizes"
TUNE_VARIANTS=-,fmai timeout 300 python scripts/tune.py 1024 2>&1 | tee -a $OUT/r02_tune_k124.txt
TUNE_VARIANTS=-,df timeout 300 python scripts/tune.py 16384 2048 2>&1 | tee -a $OUT/r02_tune_k124.txt
echo "== bench 4096-frame launches"
for v in "" fmai df dftk; do
  FSEA_BENCH_VARIANT=$v timeout 300 python bench.py --gpus 1 --steps 400 --warmup 20 --no-extra --no-cpu-baseline > $OUT/r02_bench_k_${v:-def}.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$OUT/r02_bench_k_${v:-def}.json").read().strip().splitlines()[-1])
print("variant=%-5s value %.2f Mframes/s  ms/step %.5f  launch %.5f ms  frac %.4f  kernel %s rel %.2e" % ("${v:-def}", d["value"]/1e6, d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["kernel"], d["parity_rel_l2_first_rows"]))
PY
done
