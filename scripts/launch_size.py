#!/usr/bin/env python3
"""How launch size and buffer rotation affect the per-frame cost (fixed per-launch overheads)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frequensea_amd import fsea
fsea.use_tune_library()

def dev_alloc(nbytes):
    p = ctypes.c_void_p()
    fsea._check(fsea.hip_lib().fsea_device_alloc(0, nbytes, ctypes.byref(p)))
    return p

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
TOTAL = 1 << 28     # samples: 512 MiB in, 1 GiB out
host = np.random.default_rng(1).integers(-70, 70, 2 * (1 << 24), dtype=np.int8).view(np.uint8)
d_in = dev_alloc(2 * TOTAL)
d_out = dev_alloc(4 * TOTAL)
L = fsea.hip_lib()
for off in range(0, 2 * TOTAL, host.nbytes):
    fsea._check(L.fsea_copy_to_device(0, ctypes.c_void_p(d_in.value + off), host.ctypes.data, host.nbytes))
plan = fsea.Plan(n)
plan.time_device(d_in, TOTAL // n, d_out, 3)
for frames in [TOTAL // n, TOTAL // n // 4, TOTAL // n // 16, 4096 * 8192 // n, 2048 * 8192 // n, 1024 * 8192 // n]:
    # rotate through the big buffer so that every launch touches cold data
    nslots = (TOTAL // n) // frames
    reps = max(nslots, 16)
    ev_total = 0.0
    import time
    ms_list = []
    for rnd in range(3):
        t = 0.0
        for k in range(reps):
            s = k % nslots
            t += plan.time_device(ctypes.c_void_p(d_in.value + 2 * n * frames * s), frames,
                                  ctypes.c_void_p(d_out.value + 4 * n * frames * s), 1)
        ms_list.append(t / reps)
    # back-to-back launches inside one event pair (rotation not possible through this entry point)
    same = min(plan.time_device(d_in, frames, d_out, 20) for _ in range(3))
    ms = min(ms_list)
    print("N=%d frames/launch=%-6d grid=%-4d  rotating single-launch %.4f ms (%.1f%% of 8TB/s) | same-buffer x20 %.4f ms (%.1f%%)"
          % (n, frames, plan.grid(frames)[0], ms, 6.0 * n * frames / ms / 1e6 / 80.0, same, 6.0 * n * frames / same / 1e6 / 80.0))
