#!/bin/bash
# Round 5: rocprofv3 --kernel-trace --stats of the other BASELINE sizes' launches at HEAD (the 1024- and 4096-point kernels
# carry this round's prologue balancing, the windowed 16384-point kernel its DC table in registers): one run per workload,
# 200 timed launches each, nothing else in the process.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5s; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for spec in "batch1024x32768:" "batch4096x8192:" "stft16384x8191:" "stft16384x8191:hann"; do
  wl=${spec%%:*}; win=${spec##*:}
  tag=$wl${win:+_$win}
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o bench --output-format csv -- \
    python $R/bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extra --workload $wl ${win:+--window $win} \
    > $O/bench_$tag.json 2> $O/prof_$tag.err
  for f in $(find $O/prof_$tag -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_$tag.csv; echo "== $tag"; head -3 $f; done
  python - $O/bench_$tag.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("   line: value %.4g %s, ms/step %.5f, roofline.frac %.4f (avg_launch_ms %.5f)" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("avg_launch_ms", float("nan"))))
PY
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
