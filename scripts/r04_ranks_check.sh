#!/bin/bash
# Round 4: the driver's multi-rank form of the default bench line, N = 2, 4, 8 ranks sharing the one GPU of this box over
# gloo (RCCL refuses two ranks on one device): the control path of `python -m torch.distributed.run ... bench.py --gpus N`
# (rendezvous, barriers, MAX over ranks, rank 0's single JSON line), not a scaling measurement.
mkdir -p gpurun_out
OUT=gpurun_out/r04_ranks_check.txt
: > $OUT
port=29520
for n in 2 4 8; do
  for extra in "" "--window hann"; do
    port=$((port + 1))
    echo "== $n ranks on one GPU (gloo) $extra" >> $OUT
    FSEA_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
      --master-port $port bench.py --gpus $n --steps 20 --warmup 5 $extra 2> gpurun_out/ranks_err.txt | grep '^{' | python -c "
import json, sys
lines = sys.stdin.read().strip().split('\n')
print('json lines printed:', len([l for l in lines if l]))
d = json.loads(lines[-1])
print({k: d[k] for k in ('metric', 'n_gpus', 'steps', 'value', 'ms_per_step', 'scaling', 'parity_rel_l2_first_rows')}, 'roofline.frac', d['roofline']['frac'], 'cpu_baseline' in d, 'extra' in d)
" >> $OUT 2>&1 || { echo "FAILED" >> $OUT; tail -5 gpurun_out/ranks_err.txt >> $OUT; }
  done
done
cat $OUT
