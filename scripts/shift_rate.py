#!/usr/bin/env python3
"""Rate of the frequency-shifted path (fsea_exec_u8_shifted_device, the nrf_freq_shifter fused into
the FFT kernel's load) next to the plain path, same resident data.  Host-clock timing over many
back-to-back launches (the launches are 0.2 ms each, far above launch overhead).
Usage: python scripts/shift_rate.py [N ...]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frequensea_amd import fsea  # noqa: E402

TOTAL_SAMPLES = 1 << 27


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [8192, 1024]
    L = fsea.hip_lib()
    rng = np.random.default_rng(1)
    host = rng.integers(-70, 70, 2 * TOTAL_SAMPLES, dtype=np.int8).view(np.uint8)
    d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    fsea._check(L.fsea_device_alloc(0, host.nbytes, ctypes.byref(d_in)))
    fsea._check(L.fsea_device_alloc(0, 4 * TOTAL_SAMPLES, ctypes.byref(d_out)))
    fsea._check(L.fsea_copy_to_device(0, d_in, host.ctypes.data, host.nbytes))
    for n in sizes:
        frames = TOTAL_SAMPLES // n
        plan = fsea.Plan(n, mode=fsea.MODE_MAG_NODC_F32)      # run-time-mode kernel on both sides
        for name, call in (("plain  ", lambda: plan.exec_device(d_in, frames, d_out)),
                           ("shifted", lambda: plan.exec_shifted_device(d_in, frames, d_out, 0.03, 0.25))):
            for _ in range(20):
                call()
            plan.synchronize()
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(40):
                    call()
                plan.synchronize()
                best = min(best, (time.perf_counter() - t0) / 40)
            gbs = frames * 6 * n / best / 1e9
            print("N=%-6d %s %8.3f ms  %8.1f Mframes/s  %7.1f GB/s  %5.1f%% of 8 TB/s" %
                  (n, name, best * 1e3, frames / best / 1e6, gbs, gbs / 80))
        plan.close()


if __name__ == "__main__":
    main()
