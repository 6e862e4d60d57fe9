#!/bin/bash
# Round 6 (as round 5, plus the pinned-checksum keys): the driver's own command lines, N = 1 and N = 2, 4, 8 ranks sharing the one GPU of this box over gloo (RCCL refuses
# two ranks on one device).  Since round 5 `bench.py --gpus N` runs, behind the headline's steps, the workloads north_star
# scales over a node -- the sharded fft-batch-broad sweep in both regimes and the halo-sharded 16384-point stream, each with a
# chunked gather to rank 0 -- and reports them in `extra` with the communicator's census, the checksums of what arrived against
# what the members computed, and the gather's rate.  This is the control path on one GPU, not a scaling measurement.
mkdir -p gpurun_out
OUT=gpurun_out/r06_ranks_on_one_gpu.txt
: > $OUT
summ() {
python -c "
import json, sys
lines = [l for l in sys.stdin.read().strip().split('\n') if l.startswith('{')]
print('json lines printed:', len(lines))
d = json.loads(lines[-1])
print({k: d[k] for k in ('metric', 'n_gpus', 'steps', 'value', 'ms_per_step', 'scaling')}, 'roofline.frac %.4f' % d['roofline']['frac'])
e = d.get('extra', {})
for k in ('multi_gpu_error', 'gather_backend', 'rccl_world', 'torch_world_size', 'distinct_gpus', 'gather_chunks', 'broad_sweep_ms_resident', 'broad_sweep_ms_ingest',
          'stft_stream_ms', 'broad_sweep_resident_gathered_checksum_ok', 'broad_sweep_resident_gathered_checksum', 'broad_sweep_ingest_gathered_checksum_ok',
          'broad_sweep_ingest_gathered_checksum', 'stft_stream_gathered_checksum_ok', 'stft_stream_gathered_checksum', 'broad_sweep_resident_gathered_checksum_matches_pinned', 'broad_sweep_ingest_gathered_checksum_matches_pinned', 'stft_stream_gathered_checksum_matches_pinned', 'gather_only_ms', 'gather_bytes_per_peer',
          'gather_gbps_per_link', 'gather_gbps_into_root', 'regime', 'rank_devices'):
    print('  extra.%s = %s' % (k, e.get(k)))
"
}
echo "== 1 rank (python bench.py --gpus 1 --steps 20 --warmup 5)" >> $OUT
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/ranks_err.txt > gpurun_out/r06_bench_driver_form_n1.json
cat gpurun_out/r06_bench_driver_form_n1.json | summ >> $OUT 2>&1 || { echo "FAILED" >> $OUT; tail -5 gpurun_out/ranks_err.txt >> $OUT; }
echo "  wall: $(( $(date +%s) - t0 )) s" >> $OUT
port=29620
for n in 2 4 8; do
  port=$((port + 1))
  echo "== $n ranks on one GPU (gloo): python -m torch.distributed.run --nnodes=1 --nproc-per-node $n ... bench.py --gpus $n --steps 20 --warmup 5" >> $OUT
  t0=$(date +%s)
  FSEA_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
    --master-port $port bench.py --gpus $n --steps 20 --warmup 5 2> gpurun_out/ranks_err_$n.txt > gpurun_out/r06_bench_gloo_n$n.json
  cat gpurun_out/r06_bench_gloo_n$n.json | summ >> $OUT 2>&1 || { echo "FAILED" >> $OUT; tail -15 gpurun_out/ranks_err_$n.txt >> $OUT; }
  echo "  wall: $(( $(date +%s) - t0 )) s" >> $OUT
done
cat $OUT
