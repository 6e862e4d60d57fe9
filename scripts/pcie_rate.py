#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline batch through the host-buffer entry point (never bench.py's value)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frequensea_amd import fsea
n, frames = 8192, 4096
iq = np.random.default_rng(0).integers(-70, 70, 2 * n * frames, dtype=np.int8).view(np.uint8)
plan = fsea.Plan(n)
plan.exec_host(iq, frames)
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    plan.exec_host(iq, frames)
dt = (time.perf_counter() - t0) / reps
print("fsea_exec_u8_host N=%d x %d frames (64 MiB in, 128 MiB out, pageable host memory): %.2f ms -> %.2f Mframes/s, %.1f Gsamples/s, %.1f GB/s over PCIe"
      % (n, frames, dt * 1e3, frames / dt / 1e6, frames * n / dt / 1e9, 6.0 * n * frames / dt / 1e9))
