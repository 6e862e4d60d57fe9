#!/bin/bash
# Samples rocm-smi power / clocks while the 8192-point kernel runs back to back (noise-like input, then
# constant input): evidence for the power limit discussed in DESIGN.md.  Usage: bash scripts/power_probe.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
probe() {  # label, env
  echo "== $1"
  env $2 TUNE_VARIANTS=- python - <<'PY' &
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from frequensea_amd import fsea
fsea.use_tune_library()
L = fsea.hip_lib()
n, total = 8192, 1 << 27
host = np.random.default_rng(1).integers(-70, 70, 2 * total, dtype=np.int8).view(np.uint8)
if os.environ.get("CONST_INPUT"):
    host[:] = 0x80
d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
fsea._check(L.fsea_device_alloc(0, host.nbytes, ctypes.byref(d_in)))
fsea._check(L.fsea_device_alloc(0, 4 * total, ctypes.byref(d_out)))
fsea._check(L.fsea_copy_to_device(0, d_in, host.ctypes.data, host.nbytes))
plan = fsea.Plan(n)
t_end = time.time() + 6.0
ms = []
while time.time() < t_end:
    ms.append(plan.time_device(d_in, total // n, d_out, 50))
print("kernel: median %.3f ms per launch = %.1f%% of 8 TB/s" % (np.median(ms), 6.0 * total / np.median(ms) / 1e6 / 80))
PY
  sleep 2.5
  for i in 1 2 3; do
    /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr -s ' ' | head -6
    echo "--"; sleep 0.8
  done
  wait
}
/opt/rocm/bin/rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -3
probe "noise-like input" "X=1"
probe "constant input (0x80)" "CONST_INPUT=1"
