#!/usr/bin/env python3
"""Steady-state rate of every epilogue mode (run-time-mode kernel `*_u8`, and the compile-time MAG / DB5 / DB10
kernels) on resident data.  For the two u8 pixel modes the round-2 epilogue (cast + clamp + shift/or: tuning variant
"px0") is timed alternately with the product kernel in the same process, so that the file shows what the
v_cvt_pk_u8_f32 epilogue is worth on the box it was made on.  Usage: python scripts/mode_rate.py [N ...]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frequensea_amd import fsea  # noqa: E402
fsea.use_tune_library()

TOTAL_SAMPLES = 1 << 27
NAMES = {0: "MAG_F32", 1: "DB10_U8", 2: "DB5_U8_DCFIX", 3: "COMPLEX_F32", 4: "MAG_NODC_F32", 5: "DB_F32"}
OUT_BYTES = {0: 4, 1: 1, 2: 1, 3: 8, 4: 4, 5: 4}


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1024, 4096, 8192]
    L = fsea.hip_lib()
    rng = np.random.default_rng(1)
    host = rng.integers(-70, 70, 2 * TOTAL_SAMPLES, dtype=np.int8).view(np.uint8)
    d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    fsea._check(L.fsea_device_alloc(0, host.nbytes, ctypes.byref(d_in)))
    fsea._check(L.fsea_device_alloc(0, 8 * TOTAL_SAMPLES, ctypes.byref(d_out)))
    fsea._check(L.fsea_copy_to_device(0, d_in, host.ctypes.data, host.nbytes))
    for n in sizes:
        frames = TOTAL_SAMPLES // n
        for mode in range(6):
            plans = [("", fsea.Plan(n, mode=mode))]
            if mode in (1, 2):
                try:
                    plans.append(("px0", fsea.Plan(n, mode=mode, variant="px0")))
                except fsea.FseaError:
                    pass
            times = {v: [] for v, _ in plans}
            for _, plan in plans:
                plan.time_device(d_in, frames, d_out, 20)
            for rnd in range(5):
                for v, plan in (plans if rnd % 2 == 0 else plans[::-1]):
                    times[v].append(plan.time_device(d_in, frames, d_out, 10))
            bytes_ = frames * n * (2 + OUT_BYTES[mode])
            for v, plan in plans:
                ms = sorted(times[v])[2]
                print("N=%-6d %-13s %-24s %7.3f ms %8.1f Mframes/s %8.1f Gsamples/s %7.1f GB/s %5.1f%% of 8 TB/s%s" %
                      (n, NAMES[mode], plan.kernel_name, ms, frames / ms / 1e3, frames * n / ms / 1e6,
                       bytes_ / ms / 1e6, bytes_ / ms / 1e6 / 80, "   (round-2 epilogue, same process)" if v else ""))
                plan.close()


if __name__ == "__main__":
    main()
