#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^Written\|^Composing\|^Frequency" | tail -30 | tee $OUT/r02_pytest_gpu_c.log
echo "== nrf latency, host ring vs device ring"
timeout 300 python scripts/nrf_latency.py 2>&1 | tee $OUT/r02_nrf_latency_history_modes.txt
echo "== broad sweep regimes, 1 GPU"
for regime in resident ingest; do
  timeout 300 python bench.py --workload broad --regime $regime --steps 20 --warmup 3 > $OUT/r02_broad_$regime.json 2> $OUT/r02_broad_$regime.err; cat $OUT/r02_broad_$regime.json; tail -2 $OUT/r02_broad_$regime.err
done
echo "== config 5 stream, 1 GPU"
timeout 300 python bench.py --workload stft16384stream --steps 5 --warmup 2 > $OUT/r02_stft_stream.json 2> $OUT/r02_stft_stream.err; cat $OUT/r02_stft_stream.json; tail -2 $OUT/r02_stft_stream.err
echo "== N>1 control path on one GPU (gloo, 2 ranks)"
for wl in batch8192x4096 broad stft16384stream; do
  extra=""; [ $wl = stft16384stream ] && extra="--stream-frames 4095"
  FSEA_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 3 --warmup 1 --no-extra --no-cpu-baseline --workload $wl $extra > $OUT/r02_gloo2_$wl.json 2> $OUT/r02_gloo2_$wl.err
  echo "rc=$? $(tail -c 600 $OUT/r02_gloo2_$wl.json)"; grep -i "error\|Traceback" $OUT/r02_gloo2_$wl.err | head -5
done
