#!/usr/bin/env python3
"""Joules per launch of kernel variants: package power (rocm-smi) x launch time, on noise-like input.
For every variant a worker thread launches the kernel back to back for a few seconds (steady state:
768 MiB per launch) while the main thread samples `rocm-smi --showpower --showclocks`; the launch time is
HIP-event-timed in the same run.  Usage: python scripts/energy_probe.py [variant ...]   ("-" = product kernel)
With ENERGY_CONST_INPUT=1 the input is constant bytes (nothing toggles)."""
import ctypes, os, re, subprocess, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frequensea_amd import fsea
fsea.use_tune_library()

N = int(os.environ.get("ENERGY_N", "8192"))
TOTAL = 1 << 27
MODE = int(os.environ.get("ENERGY_MODE", "0"))          # epilogue mode (0 = MAG_F32, 1 = DB10_U8, 2 = DB5_U8_DCFIX, ...)
OUT_BYTES = {0: 4, 1: 1, 2: 1, 3: 8, 4: 4, 5: 4}[MODE]
SECONDS = float(os.environ.get("ENERGY_SECONDS", "4"))
variants = sys.argv[1:] or ["-", "nd", "abl_io", "abl_nolds", "abl_noflop"]
L = fsea.hip_lib()
host = np.random.default_rng(1).integers(-70, 70, 2 * TOTAL, dtype=np.int8).view(np.uint8)
if os.environ.get("ENERGY_CONST_INPUT"):
    host[:] = 0x80
d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
fsea._check(L.fsea_device_alloc(0, host.nbytes, ctypes.byref(d_in)))
fsea._check(L.fsea_device_alloc(0, 4 * TOTAL, ctypes.byref(d_out)))
fsea._check(L.fsea_copy_to_device(0, d_in, host.ctypes.data, host.nbytes))
frames = TOTAL // N


def smi():
    out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    w = re.search(r"Package Power \(W\):\s*([0-9.]+)", out)
    s = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", out)
    return (float(w.group(1)) if w else float("nan")), (float(s.group(1)) if s else float("nan"))


print("input: %s, N=%d, mode %d, %d frames per launch (%.0f MiB in + %.0f MiB out)" %
      ("constant 0x80" if os.environ.get("ENERGY_CONST_INPUT") else "noise-like int8", N, MODE, frames, 2 * TOTAL / 2**20,
       OUT_BYTES * TOTAL / 2**20))
print("%-12s %-28s %9s %8s %8s %10s %10s %8s" % ("variant", "kernel", "ms/launch", "W", "sclk MHz", "J/launch", "uJ/frame", "% 8TB/s"))
for var in variants:
    plan = fsea.Plan(N, variant="" if var == "-" else var, mode=MODE)
    ms_list, stop = [], False

    def worker():
        while not stop:
            ms_list.append(plan.time_device(d_in, frames, d_out, 25))
    th = threading.Thread(target=worker)
    th.start()
    time.sleep(1.2)                                  # clocks and power settle
    samples = []
    t_end = time.time() + SECONDS
    while time.time() < t_end:
        samples.append(smi())
        time.sleep(0.25)
    stop = True
    th.join()
    ms = float(np.median(ms_list[len(ms_list) // 3:]))
    w = float(np.nanmedian([a for a, _ in samples]))
    clk = float(np.nanmedian([b for _, b in samples]))
    print("%-12s %-28s %9.4f %8.0f %8.0f %10.4f %10.3f %8.1f" %
          (var, plan.kernel_name, ms, w, clk, w * ms * 1e-3, w * ms * 1e-3 / frames * 1e6, (2.0 + OUT_BYTES) * TOTAL / ms / 1e6 / 80.0))
    plan.close()
