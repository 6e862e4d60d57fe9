#!/bin/bash
# First contact with a multi-GPU node (this round's work never saw one): everything that involves more than one GPU,
# in one go.  Usage: bash scripts/multi_gpu_check.sh [N]   (N = GPUs to use, default: all)
#   1. the C tool over RCCL: fsea-fft-sweep --devices 0-(N-1) against fsea-fft-batch + fsea-fft-stitch (pytest, -k all)
#   2. bench.py at 1, 2, 4 ... N ranks: the default workload (frames sharded, no collective), the fft-batch-broad sweep in
#      both regimes (chunked gather over RCCL) and the halo-sharded 16384-point stream
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
N=${1:-$(python -c "from frequensea_amd import fsea; print(fsea.device_count())")}
echo "== GPUs: $N"
python -m pytest tests/test_gpu_tools.py -q -m gpu -k "multi_member" 2>&1 | tail -3
port=29600
n=1
while [ $n -le $N ]; do
  for wl in "" "--workload broad --regime resident" "--workload broad --regime ingest" "--workload stft16384stream"; do
    port=$((port + 1))
    if [ $n -eq 1 ]; then
      python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline $wl 2>/dev/null | tail -1 | cut -c1-420
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
        bench.py --gpus $n --steps 20 --warmup 5 $wl 2>/dev/null | grep "^{" | cut -c1-420
    fi
  done
  n=$((n * 2))
done
