#!/usr/bin/env python3
"""Round 6 ablation (new: nobody had varied the MEMORY TYPE of the caller's buffers): the headline kernel, its I/O skeleton and
a plain 1 : 2 stream on buffers from hipMalloc (default: cached in L2 and the Infinity Cache) against buffers from
hipExtMallocWithFlags(hipDeviceMallocUncached / hipDeviceMallocFinegrained / hipDeviceMallocContiguous), input and output
varied separately.  Same launch shape as bench.py's timed steps (4096 frames of 8192 points, six rotating sets, HIP events).
70 % of the joules per frame are data movement, and both directions are pure streams: if the memory type
moves the rate by >= 3 %, the library's allocator (fsea_device_alloc) can offer it.  Prints one JSON line per configuration,
the configurations interleaved round by round so that clock drift hits them alike."""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frequensea_amd import fsea  # noqa: E402
fsea.use_tune_library()

N, FRAMES, SETS, PEAK = 8192, 4096, 6, 8000.0
FLAGS = {"default": None, "uncached": 0x3, "finegrained": 0x1, "contiguous": 0x4}


def main():
    L = fsea.hip_lib()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    in_bytes, out_bytes = 2 * N * FRAMES, 4 * N * FRAMES
    host = np.random.default_rng(3).normal(0, 20, in_bytes).round().clip(-128, 127).astype(np.int8).view(np.uint8)

    def alloc(kind, size):
        p = ctypes.c_void_p()
        if FLAGS[kind] is None:
            fsea._check(L.fsea_device_alloc(0, size, ctypes.byref(p)))
        else:
            rc = hip.hipExtMallocWithFlags(ctypes.byref(p), size, FLAGS[kind])
            if rc != 0:
                return None
        return p

    configs = [("default", "default"), ("uncached", "uncached"), ("default", "uncached"), ("uncached", "default"),
               ("finegrained", "finegrained"), ("contiguous", "contiguous")]
    bufs = {}
    for cin, cout in configs:
        ins, outs, ok = [], [], True
        for s in range(SETS):
            a, b = alloc(cin, in_bytes), alloc(cout, out_bytes)
            if a is None or b is None:
                ok = False
                break
            rolled = np.roll(host, 16 * s)
            fsea._check(L.fsea_copy_to_device(0, a, rolled.ctypes.data, in_bytes))
            ins.append(a)
            outs.append(b)
        if ok:
            bufs[(cin, cout)] = (ins, outs)
        else:
            print(json.dumps({"in": cin, "out": cout, "error": "allocation refused"}))
    alg = in_bytes + out_bytes
    plans = {"product": fsea.Plan(N, variant=""), "io_skeleton": fsea.Plan(N, variant="abl_io_nt")}
    vp = ctypes.c_void_p
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.5:
        for ins, outs in bufs.values():
            plans["product"].time_rotating(ins, FRAMES, outs, 5 * SETS)
    rounds = {(cfg, k): [] for cfg in bufs for k in ("product", "io_skeleton", "copy")}
    for r in range(7):
        for cfg, (ins, outs) in bufs.items():
            for k, p in plans.items():
                rounds[(cfg, k)].append(p.time_rotating(ins, FRAMES, outs, 20 * SETS))
            a_in = (vp * SETS)(*[p.value for p in ins])
            a_out = (vp * SETS)(*[p.value for p in outs])
            ms = ctypes.c_float(0)
            fsea._check(L.fsea_tune_stream_1to2(a_in, a_out, SETS, in_bytes, 0, None, 20 * SETS, ctypes.byref(ms)))
            rounds[(cfg, "copy")].append(ms.value)
    # rows of the product kernel must not depend on where the buffers live
    ref = None
    for cfg, (ins, outs) in bufs.items():
        plans["product"].exec_device(ins[0], FRAMES, outs[0])
        plans["product"].synchronize()
        row = np.empty(4 * N, np.float32)
        fsea._check(L.fsea_copy_to_host(0, row.ctypes.data, outs[0], row.nbytes))
        if ref is None:
            ref = row
        assert np.array_equal(ref, row), cfg
    for cfg in bufs:
        line = {"in": cfg[0], "out": cfg[1]}
        for k in ("product", "io_skeleton", "copy"):
            ms = float(np.median(rounds[(cfg, k)]))
            line[k + "_launch_us"] = round(1e3 * ms, 2)
            line[k + "_frac"] = round(alg / (ms * 1e-3) / 1e9 / PEAK, 4)
        print(json.dumps(line))


if __name__ == "__main__":
    main()
