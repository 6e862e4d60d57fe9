#!/usr/bin/env python3
"""What bounds the headline kernel from above, measured on THIS box in the streaming regime bench.py uses (4096 frames of
8192 points per launch, six rotating buffer sets, HIP events on the launch stream): the product kernel, its I/O skeleton
(the same loads, conversion-free pass-through and row stores with the product's cache policy; no LDS exchange, no
butterflies: tuning variant abl_io_nt, wrong rows by design) and a plain 1 : 2 read/write stream.  Prints ONE JSON line;
bench.py runs this as a subprocess (the tuning library is never loaded into the process that measures `value`)."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frequensea_amd import fsea  # noqa: E402
fsea.use_tune_library()

N, FRAMES, SETS, PEAK = 8192, 4096, 6, 8000.0


def main():
    L = fsea.hip_lib()
    in_bytes, out_bytes = 2 * N * FRAMES, 4 * N * FRAMES
    host = np.random.default_rng(3).normal(0, 20, in_bytes).round().clip(-128, 127).astype(np.int8).view(np.uint8)
    ins, outs = [], []
    for s in range(SETS):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        fsea._check(L.fsea_device_alloc(0, in_bytes, ctypes.byref(a)))
        fsea._check(L.fsea_device_alloc(0, out_bytes, ctypes.byref(b)))
        rolled = np.roll(host, 16 * s)                 # (kept alive across the call)
        fsea._check(L.fsea_copy_to_device(0, a, rolled.ctypes.data, in_bytes))
        ins.append(a)
        outs.append(b)
    alg = in_bytes + out_bytes
    res = {"frames_per_launch": FRAMES, "fft_size": N, "buffer_sets": SETS, "algorithmic_bytes_per_launch": alg}
    plans = {"product": fsea.Plan(N, variant=""), "io_skeleton": fsea.Plan(N, variant="abl_io_nt")}
    import time
    t_pre = time.perf_counter()                    # clocks and caches: the same 0.25 s pre-warm as bench.py's headline
    while time.perf_counter() - t_pre < 0.25:
        for p in plans.values():
            p.time_rotating(ins, FRAMES, outs, 5 * SETS)
    rounds = {k: [] for k in list(plans) + ["copy"]}
    vp = ctypes.c_void_p
    a_in = (vp * SETS)(*[p.value for p in ins])
    a_out = (vp * SETS)(*[p.value for p in outs])
    for r in range(5):
        for k, p in plans.items():
            rounds[k].append(p.time_rotating(ins, FRAMES, outs, 20 * SETS))
        ms = ctypes.c_float(0)
        fsea._check(L.fsea_tune_stream_1to2(a_in, a_out, SETS, in_bytes, 0, None, 20 * SETS, ctypes.byref(ms)))
        rounds["copy"].append(ms.value)
    for k, v in rounds.items():
        ms = float(np.median(v))
        res[k + "_launch_ms"] = ms
        res[k + "_frac"] = alg / (ms * 1e-3) / 1e9 / PEAK
    res["kernels"] = {k: p.kernel_name for k, p in plans.items()}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
