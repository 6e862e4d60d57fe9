#!/bin/bash
# Config 5's half-overlap kernel: frames per run 8 (round 3's default) against 16 -- rate (alternating bench runs) and HBM fetch
# traffic (FETCH_SIZE, rocprofv3 PMC pass, kernel-trace only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4s; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for m in 16 32 64; do
  echo -n "run_max $m: "; FSEA_HALF_RUN_MAX=$m python $R/bench.py --workload stft16384x8191 --steps 300 --warmup 30 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], 'frames/s %.4g' % d['value_events'], 'ms/launch %.4f' % d['roofline']['avg_launch_ms'], 'frac %.4f' % d['roofline']['frac'])"
done; done
for m in 32; do for c in FETCH_SIZE; do
  FSEA_HALF_RUN_MAX=$m timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$m_$c -o pmc --output-format csv -- python $R/bench.py --workload stft16384x8191 --steps 20 --warmup 4 --no-extra --no-cpu-baseline > /dev/null 2> $O/pmc.err
  f=$(find $O/pmc_$m_$c -name "*counter_collection.csv" | head -1)
  echo -n "run_max $m: "; python3 $R/scripts/pmc_summary.py $f; rm -rf $O/pmc_$m_$c
done; done
