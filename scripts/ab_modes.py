#!/usr/bin/env python3
"""Every (size, epilogue mode) through several builds of libfsea_hip.so in ONE process (raw int8 input, 2^27 samples per
launch, resident): rounds of launches alternate between the libraries after a common pre-warm; HIP-event time per launch,
median of the rounds.  The last column compares the current library with the first one given (e.g. round 2's).
Usage: python scripts/ab_modes.py OLD.so [OLD2.so ...] [-- N ...]   (the current product library is added last)"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
sizes = [64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384]
if "--" in args:
    sizes = [int(a) for a in args[args.index("--") + 1:]]
    args = args[:args.index("--")]
paths = args + [os.path.join(ROOT, "frequensea_amd", "libfsea_hip.so")]
vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
TOTAL = 1 << 27
NAMES = {0: "MAG_F32", 1: "DB10_U8", 2: "DB5_U8_DCFIX", 3: "COMPLEX_F32", 4: "MAG_NODC_F32", 5: "DB_F32"}
OUT_BYTES = {0: 4, 1: 1, 2: 1, 3: 8, 4: 4, 5: 4}
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
libs = []
for path in paths:
    L = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    L.fsea_plan_create.argtypes = [ctypes.POINTER(vp), ci, ci, ci, ci]
    L.fsea_plan_destroy.argtypes = [vp]
    L.fsea_exec_u8_device.argtypes = [vp, vp, sz, ci, vp, vp]
    L.fsea_plan_kernel_name.argtypes = [vp]
    L.fsea_plan_kernel_name.restype = ctypes.c_char_p
    libs.append((os.path.basename(path), L))
host = np.random.default_rng(1).integers(-70, 70, 2 * TOTAL, dtype=np.int8)
d_in = torch.from_numpy(host).to(dev)
d_out = torch.empty(8 * TOTAL, dtype=torch.uint8, device=dev)
t0 = time.perf_counter()
warm = vp()
assert libs[-1][1].fsea_plan_create(ctypes.byref(warm), 8192, 8192, 0, 0) == 0
while time.perf_counter() - t0 < 0.5:
    for _ in range(16):
        libs[-1][1].fsea_exec_u8_device(warm, d_in.data_ptr(), TOTAL // 8192, 1, d_out.data_ptr(), stream)
    torch.cuda.synchronize()
libs[-1][1].fsea_plan_destroy(warm)
print("libraries: " + ", ".join(n for n, _ in libs) + "; ms per launch of 2^27 samples, median of 7 interleaved rounds of 10 launches")
worst = (9.0, None)
for n in sizes:
    frames = TOTAL // n
    for mode in range(6):
        plans = []
        for name, L in libs:
            p = vp()
            if L.fsea_plan_create(ctypes.byref(p), n, n, mode, 0) != 0:
                plans.append((name, L, None))
                continue
            plans.append((name, L, p))
        ms = {name: [] for name, _, _ in plans}
        for rnd in range(8):
            for name, L, p in (plans if rnd % 2 == 0 else plans[::-1]):
                if p is None:
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    L.fsea_exec_u8_device(p, d_in.data_ptr(), frames, 1, d_out.data_ptr(), stream)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    ms[name].append(e0.elapsed_time(e1) / 10)
        med = {k: (float(np.median(v)) if v else float("nan")) for k, v in ms.items()}
        cur, old = med[plans[-1][0]], med[plans[0][0]]
        kname = plans[-1][1].fsea_plan_kernel_name(plans[-1][2]).decode()
        ratio = old / cur
        if ratio < worst[0]:
            worst = (ratio, (n, NAMES[mode]))
        print("N=%-6d %-13s %-26s " % (n, NAMES[mode], kname) + "  ".join("%s %.4f" % (k.replace("libfsea_hip", "lib").replace(".so", ""), v) for k, v in med.items()) +
              "   %5.1f %% of 8 TB/s   x%.3f vs %s" % ((2 + OUT_BYTES[mode]) * TOTAL / cur / 1e6 / 80, ratio, plans[0][0]))
        for name, L, p in plans:
            if p is not None:
                L.fsea_plan_destroy(p)
print("lowest current/old rate ratio: x%.3f at %s" % worst)
