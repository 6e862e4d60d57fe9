#!/bin/bash
# Round 6, one GPU-box visit: the whole GPU tier (with durations), smoke, the bench line in the driver's form (three times),
# rocprofv3 --kernel-trace --stats of the headline launches, of config 4's sweep and of config 5's STFT, PMC passes of the
# headline kernel (separate runs, kernel-trace only), and the driver's command lines at 1 / 2 / 4 / 8 ranks on the one GPU (gloo).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6f; mkdir -p $O
cd $R; export TMPDIR=/tmp
echo "== pytest -m gpu"
t0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=12 > $O/pytest_gpu_full.log 2>&1
grep -E "passed|failed|rror" $O/pytest_gpu_full.log | tail -4 | tee $O/pytest_gpu.log
echo "pytest -m gpu wall: $(( $(date +%s) - t0 )) s" | tee -a $O/pytest_gpu.log
grep -A14 "slowest" $O/pytest_gpu_full.log | tee -a $O/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -4 | tee $O/smoke.log
for run in a b c; do
  echo "== bench (driver form) $run"
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form_$run.json 2> $O/bench_driver_form_$run.err
  python - $O/bench_driver_form_$run.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r, e = d["roofline"], d["extra"]
print("value %.4g  value_events %.4g  ms/step %.5f  frac %.4f  official_over_median %.4f (events %.4f)  prewarm %.3f s  regions %s" %
      (d["value"], d["value_events"], d["ms_per_step"], r["frac"], d["official_over_median"], d["official_over_median_events"],
       d["config"]["clock_prewarm_s"], {k: round(v, 5) for k, v in d["headline_regions_ms"].items()}))
print("  kernel_over_io_skeleton %.3f  valu_issue %.3f  lds_active %.3f  W %s / %s  frac_long %.4f" %
      (r.get("kernel_over_io_skeleton", float("nan")), r.get("valu_issue_frac", float("nan")), r.get("lds_active_frac", float("nan")),
       r.get("package_power_w_in_bench_shape"), r.get("package_power_w_long_launches"), r.get("frac_long_launches", float("nan"))))
for k in ("hann_n8192_over_rect", "stft16384_hann_over_rect", "stft16384_roofline_frac", "broad_sweep_1gpu_ms", "broad_sweep_roofline_frac",
          "broad_sweep_ms_ingest", "stft_stream_ms", "two_stream_frames_per_sec_n8192", "roofline_frac_n1024", "multi_gpu_error",
          "broad_sweep_resident_gathered_checksum_matches_pinned", "stft_stream_gathered_checksum_matches_pinned"):
    print("  %s = %s" % (k, e.get(k)))
print("  multi_gpu_lines.stft_stream.kernel_roofline_frac = %s" % e["multi_gpu_lines"]["stft_stream"]["kernel_roofline_frac"])
print("  cpu_baseline: %.4g frames/s on %d cores (%s)" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"]))
PY
done
cd /tmp
for wl in batch8192x4096 broad stft16384x8191 batch1024x32768; do
  echo "== rocprofv3 kernel stats: $wl"
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$wl -o bench --output-format csv -- \
    python $R/bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extra --workload $wl > $O/prof_$wl.json 2> $O/prof_$wl.err
  for f in $(find $O/prof_$wl -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_$wl.csv; head -4 $f; done
  head -c 400 $O/prof_$wl.json; echo
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
cd $R
echo "== PMC passes: headline kernel"
bash scripts/pmc.sh r06 > $O/pmc.log 2>&1
cat gpurun_out/pmc_r06/*.summary.txt > $O/pmc_fsea_fft8192_u8_mag.txt 2>/dev/null; tail -4 $O/pmc.log
rm -rf gpurun_out/pmc_r06/*/
echo "== the driver's command lines, 1 / 2 / 4 / 8 ranks on this GPU"
bash scripts/r06_ranks_check.sh > $O/ranks.log 2>&1; grep -E "^==|multi_gpu_error|checksum|wall|value" gpurun_out/r06_ranks_on_one_gpu.txt | cut -c1-220
