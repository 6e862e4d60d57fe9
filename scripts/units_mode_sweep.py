#!/usr/bin/env python3
"""Static interleave vs ticket pools by launch length: two plans with the distribution pinned either way
(fsea_plan_set_unit_distribution).  Rotating buffer sets (streaming regime), interleaved rounds.
Usage: python scripts/units_mode_sweep.py N [N ...]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frequensea_amd import fsea  # noqa: E402
fsea.use_tune_library()
L = fsea.hip_lib()
SETS, ROUNDS = 6, 9
MODE = int(os.environ.get("SWEEP_MODE", "0"))
OUT_BYTES = {0: 4, 1: 1, 2: 1}[MODE]


def dev_alloc(nbytes):
    p = ctypes.c_void_p()
    fsea._check(L.fsea_device_alloc(0, nbytes, ctypes.byref(p)))
    return p


for n in [int(a) for a in sys.argv[1:]] or [8192]:
    max_frames = (1 << 28) // n // 2
    host = np.random.default_rng(1).integers(-70, 70, 2 * n * max_frames, dtype=np.int8).view(np.uint8)
    d_ins = [dev_alloc(host.nbytes) for _ in range(SETS)]
    d_outs = [dev_alloc(OUT_BYTES * n * max_frames) for _ in range(SETS)]
    for d in d_ins:
        fsea._check(L.fsea_copy_to_device(0, d, host.ctypes.data, host.nbytes))
    plans = {}
    for name, policy in (("tickets", fsea.UNITS_TICKETS), ("static", fsea.UNITS_STATIC)):
        plans[name] = fsea.Plan(n, mode=MODE)
        plans[name].set_unit_distribution(policy)
    grid = plans["static"].grid(max_frames)[0]
    frames = 256
    while frames <= max_frames:
        res = {k: [] for k in plans}
        reps = max(20, min(400, (1 << 22) // frames))
        for r in range(ROUNDS):
            for name in (list(plans) if r % 2 == 0 else list(plans)[::-1]):
                res[name].append(plans[name].time_rotating(d_ins, frames, d_outs, reps))
        t = {k: float(np.median(v)) for k, v in res.items()}
        per_wg = frames / (grid * (2 if n == 4096 else 1))
        print("N=%-6d frames/launch %-6d units/workgroup %-6.1f  tickets %8.2f us  static %8.2f us  static/tickets %.3f   (%.1f %% / %.1f %% of 8 TB/s)" %
              (n, frames, per_wg, 1e3 * t["tickets"], 1e3 * t["static"], t["static"] / t["tickets"],
               (2 + OUT_BYTES) * n * frames / (t["tickets"] * 1e-3) / 8e12 * 100, (2 + OUT_BYTES) * n * frames / (t["static"] * 1e-3) / 8e12 * 100))
        frames *= 2
    for p in plans.values():
        p.close()
    for d in d_ins + d_outs:
        L.fsea_device_free(0, d)
