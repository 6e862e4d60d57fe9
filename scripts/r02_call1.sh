#!/bin/bash
# Round-2 GPU visit 1: V2 schedule (intra-wave first exchange, two barriers) against the round-1 kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
/opt/rocm/bin/rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -4
echo "== tune 8192: default vs v2 (768 MiB per launch, interleaved rounds)"
TUNE_VARIANTS=-,v2,abl_io timeout 300 python scripts/tune.py 8192 2>&1 | tee $OUT/r02_tune_v2.txt
echo "== bench, 4096-frame launches"
for v in "" v2; do
  FSEA_BENCH_VARIANT=$v timeout 300 python bench.py --gpus 1 --steps 400 --warmup 20 --no-extra --no-cpu-baseline > $OUT/r02_bench_${v:-v1}.json 2> $OUT/r02_bench_${v:-v1}.err
  python - <<PY
import json
d=json.loads(open("$OUT/r02_bench_${v:-v1}.json").read().strip().splitlines()[-1])
print("variant=%-3s value %.2f Mframes/s  ms/step %.5f  launch %.5f ms  frac %.4f  kernel %s rel %.2e" % ("${v:-v1}", d["value"]/1e6, d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["kernel"], d["parity_rel_l2_first_rows"]))
PY
done
echo "== workgroup traces"
for v in "" v2; do
  FSEA_VARIANT=$v timeout 120 python scripts/wg_trace.py 8192 4096 2>&1 | tee $OUT/r02_wg_trace_${v:-v1}.txt
done
echo "== second round of the bench (order reversed)"
for v in v2 ""; do
  FSEA_BENCH_VARIANT=$v timeout 300 python bench.py --gpus 1 --steps 400 --warmup 20 --no-extra --no-cpu-baseline > $OUT/r02_bench_${v:-v1}_b.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$OUT/r02_bench_${v:-v1}_b.json").read().strip().splitlines()[-1])
print("variant=%-3s value %.2f Mframes/s  ms/step %.5f  launch %.5f ms  frac %.4f" % ("${v:-v1}", d["value"]/1e6, d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
PY
done
echo "== gpu tests (quick subset)"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "8192 or full_size" 2>&1 | tail -4
