#!/bin/bash
# Round 5: the whole GPU tier, smoke, the bench line in the driver's form (twice), rocprofv3 stats of the Hann-windowed launches.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5f; mkdir -p $O
cd $R; export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror" | tail -4 | tee $O/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -4 | tee $O/smoke.log
for run in a b; do
  echo "== bench (driver form) $run"
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form_$run.json 2> $O/bench_driver_form_$run.err
  python - $O/bench_driver_form_$run.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r, e = d["roofline"], d["extra"]
print("value %.4g  value_events %.4g  ms/step %.5f  frac %.4f  frac_by_step_time %.4f  kernel_over_io_skeleton %.3f  valu_issue %.3f  lds_active %.3f  W %s / %s  frac_long %.4f" %
      (d["value"], d["value_events"], d["ms_per_step"], r["frac"], r["frac_by_step_time"], r.get("kernel_over_io_skeleton", float("nan")),
       r.get("valu_issue_frac", float("nan")), r.get("lds_active_frac", float("nan")), r.get("package_power_w_in_bench_shape"), r.get("package_power_w_long_launches"),
       r.get("frac_long_launches", float("nan"))))
for k in ("hann_n8192_over_rect", "stft16384_hann_over_rect", "stft16384_roofline_frac", "stft16384_hann_roofline_frac", "broad_sweep_1gpu_ms",
          "broad_sweep_ms_ingest", "stft_stream_ms", "two_stream_frames_per_sec_n8192", "roofline_frac_n1024", "multi_gpu_error",
          "energy_uj_per_frame_n8192_rect", "energy_uj_per_frame_n8192_hann", "energy_uj_per_frame_n8192_rect_bench_shape"):
    print("  %s = %s" % (k, e.get(k)))
print("  cpu_baseline: %.4g frames/s on %d cores (%s)" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"]))
PY
done
echo "== rocprofv3 kernel stats, Hann-windowed headline launches"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_w -o bench --output-format csv -- \
  python $R/bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extra --window hann > $O/prof_bench_hann.json 2> $O/prof_w.err
for f in $(find $O/prof_w -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_bench_hann.csv; head -3 $f; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
