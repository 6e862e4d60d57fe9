#!/bin/bash
# Round 4: the bench line in the driver's form (twice: reproducibility of the extras), the contract test, the stream workload with a taper.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4b; mkdir -p $O
cd $R; export TMPDIR=/tmp
for i in 1 2; do
  echo "== bench (driver form) run $i"
  ( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_form_$i.json 2> $O/bench_driver_form_$i.err
  cat $O/bench_driver_form_$i.json; tail -4 $O/bench_driver_form_$i.err
done
echo "== stream with a Hann taper"
python bench.py --workload stft16384stream --window hann --steps 20 --warmup 3 2>&1 | tail -2 | tee $O/bench_stream_hann.json
python bench.py --workload stft16384stream --steps 20 --warmup 3 2>&1 | tail -2 | tee $O/bench_stream_rect.json
echo "== contract test"
timeout 900 python -m pytest tests/test_bench_contract.py -x -q -m gpu 2>&1 | tail -5
