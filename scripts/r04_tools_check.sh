#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e; mkdir -p $O
cd $R; export TMPDIR=/tmp
echo "== tool tests + distributed single-GPU paths"
timeout 1500 python -m pytest tests/test_gpu_tools.py tests/test_gpu_parity.py -x -q -k "tool or sweep or stitch or batch or capture or graph or slot or stream" 2>&1 | tail -4 | tee $O/pytest_tools.log
echo "== bench broad: resident, ingest (chunks alternate on two streams)"
python bench.py --workload broad --steps 20 --warmup 3 2>/dev/null | tail -1 | tee $O/broad_resident.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step','ms_per_step_kernel_events','ms_per_step_two_streams','host_issue_ms_per_step')})"
python bench.py --workload broad --regime ingest --chunks 8 --steps 5 --warmup 2 2>/dev/null | tail -1 | tee $O/broad_ingest.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step','stitched_pixel_max_diff_vs_numpy_guard')})"
echo "== two ranks on one GPU over gloo (chunked gather, lanes)"
FSEA_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload broad --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('n_gpus','value','ms_per_step','stitched_pixel_max_diff_vs_numpy_guard')})"
FSEA_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --workload stft16384stream --window hann --stream-frames 4095 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('n_gpus','value','ms_per_step','parity_rel_l2_first_rows')})"
