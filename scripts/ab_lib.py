#!/usr/bin/env python3
"""A/B of two builds of libfsea_hip.so in ONE process on the headline launch (8192-pt x 4096 frames, six rotating
buffer sets, one stream): rounds of back-to-back launches alternate between the libraries, torch events on the launch
stream give the mean launch time per round.  Usage: python scripts/ab_lib.py OLD.so [NEW.so] [N] [FRAMES]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
old = sys.argv[1]
new = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "frequensea_amd", "libfsea_hip.so")
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
sets, launches, rounds = 6, 200, 12

libs = []
for path in (old, new):
    L = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    L.fsea_plan_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.fsea_exec_u8_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    p = ctypes.c_void_p()
    assert L.fsea_plan_create(ctypes.byref(p), n, n, 0, 0) == 0
    libs.append((os.path.basename(path), L, p))

dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
host = rng.integers(-70, 70, 2 * frames * n, dtype=np.int8)
ins = [torch.from_numpy(np.roll(host, 16 * s)).to(dev) for s in range(sets)]
outs = [torch.empty(frames * n, dtype=torch.float32, device=dev) for _ in range(sets)]
stream = torch.cuda.current_stream().cuda_stream
res = {name: [] for name, _, _ in libs}
for rnd in range(rounds + 2):
    for name, L, p in (libs if rnd % 2 == 0 else libs[::-1]):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(launches):
            assert L.fsea_exec_u8_device(p, ins[k % sets].data_ptr(), frames, 1, outs[k % sets].data_ptr(), stream) == 0
        e1.record()
        torch.cuda.synchronize()
        if rnd >= 2:
            res[name].append(1e3 * e0.elapsed_time(e1) / launches)
ref = None
for name, _, _ in libs:
    v = np.array(res[name])
    print("%-32s N=%d frames=%d  us/launch: median %.2f  mean %.2f  min %.2f  max %.2f   (%.1f %% of 8 TB/s at the median)" %
          (name, n, frames, np.median(v), v.mean(), v.min(), v.max(), 6.0 * n * frames / (np.median(v) * 1e-6) / 8e12 * 100))
