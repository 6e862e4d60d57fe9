#!/bin/bash
# Round 4, last visit: the whole GPU tier, smoke, the bench line in the driver's form.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4f; mkdir -p $O
cd $R; export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror" | tail -4 | tee $O/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -4 | tee $O/smoke.log
echo "== bench (driver form)"
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4f/bench_driver_form.json"))
r, e = d["roofline"], d["extra"]
print("value %.4g  value_events %.4g  ms/step %.5f  frac %.4f  frac_by_step_time %.4f  kernel_over_io_skeleton %.3f" %
      (d["value"], d["value_events"], d["ms_per_step"], r["frac"], r["frac_by_step_time"], r.get("kernel_over_io_skeleton", float("nan"))))
for k in ("hann_n8192_over_rect", "stft16384_hann_over_rect", "stft16384_roofline_frac", "stft16384_hann_roofline_frac", "broad_sweep_1gpu_ms",
          "broad_sweep_1gpu_kernel_ms", "broad_sweep_1gpu_ms_two_streams", "two_stream_frames_per_sec_n8192", "roofline_frac_n1024",
          "energy_uj_per_frame_n8192_rect", "energy_uj_per_frame_n8192_hann"):
    print("  %s = %s" % (k, e.get(k)))
print("  cpu_baseline: %.4g frames/s on %d cores (%s)" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"]))
PY
