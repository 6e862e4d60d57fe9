#!/usr/bin/env python3
"""The windowed headline launch (8192 points x 4096 frames, Hann, six rotating buffer sets -- bench.py's launch shape) and
its rectangular twin through several builds of libfsea_hip.so in ONE process: interleaved rounds after a common pre-warm,
HIP-event time per launch.  Usage: python scripts/ab_window.py LIB.so [LIB.so ...] (the current product library is added last)"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
paths = sys.argv[1:] + [os.path.join(ROOT, "frequensea_amd", "libfsea_hip.so")]
vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
n = int(os.environ.get("AB_N", "8192"))                 # AB_N: another transform size, same 64 MiB of samples per launch
hop = int(os.environ.get("AB_HOP", str(n)))             # AB_HOP = n / 2: BASELINE config 5's overlapped frames
frames, sets = (1 << 25) // hop - (1 if hop < n else 0), 6
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
host = np.random.default_rng(1).integers(-70, 70, 2 * ((frames - 1) * hop + n), dtype=np.int8)
ins = [torch.from_numpy(np.roll(host, 16 * s)).to(dev) for s in range(sets)]
outs = [torch.empty(frames * n, dtype=torch.float32, device=dev) for _ in range(sets)]
w = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)).astype(np.float32)
plans = []
for path in paths:
    L = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    L.fsea_plan_create.argtypes = [ctypes.POINTER(vp), ci, ci, ci, ci]
    L.fsea_exec_u8_device.argtypes = [vp, vp, sz, ci, vp, vp]
    L.fsea_plan_set_window.argtypes = [vp, vp]
    for tag in ("rect", "hann"):
        p = vp()
        assert L.fsea_plan_create(ctypes.byref(p), n, hop, 0, 0) == 0
        if tag == "hann":
            assert L.fsea_plan_set_window(p, w.ctypes.data) == 0
        plans.append((os.path.basename(path) + ":" + tag, L, p))


def run(L, p, count):
    for k in range(count):
        L.fsea_exec_u8_device(p, ins[k % sets].data_ptr(), frames, 1, outs[k % sets].data_ptr(), stream)


t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    for _, L, p in plans:
        run(L, p, 32)
    torch.cuda.synchronize()
res = {name: [] for name, _, _ in plans}
# AB_REGION launches per timed region (default 200); bench.py's informational figures use regions of 20 launches, each
# started from an idle chip behind a device synchronise -- AB_REGION=20 AB_ROUNDS=60 measures in that form
REGION = int(os.environ.get("AB_REGION", "200"))
ROUNDS = int(os.environ.get("AB_ROUNDS", "15"))
print("N = %d, hop %d, %d frames per launch; timed regions of %d launches, %d interleaved rounds" % (n, hop, frames, REGION, ROUNDS))
for rnd in range(ROUNDS):
    for name, L, p in (plans if rnd % 2 == 0 else plans[::-1]):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(L, p, REGION)
        e1.record()
        torch.cuda.synchronize()
        res[name].append(1e3 * e0.elapsed_time(e1) / REGION)
for name, _, _ in plans:
    v = np.array(res[name])
    print("%-44s us/launch: median %.2f  min %.2f  max %.2f" % (name, np.median(v), v.min(), v.max()))
