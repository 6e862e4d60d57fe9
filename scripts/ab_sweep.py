#!/usr/bin/env python3
"""Config 4 (fft-batch-broad sweep on one GPU: 512 tiles x 256 frames x 4096 points, DB5 pixels written straight into the
stitched image, fsea_exec_u8_tiled_device) through several builds of libfsea_hip.so in ONE process: rounds of sweeps
alternate between the libraries after a common clock pre-warm; per round the HIP-event time per sweep and the host wall
clock per sweep (K = 20, the driver's form).  Also the headline launch (8192 x 4096 MAG) the same way.
Usage: python scripts/ab_sweep.py LIB.so [LIB.so ...]   (the current product library is always added last)
Old builds: git archive <round-end commit> frequensea_amd/csrc include | tar -x -C /tmp/rNN; make -C ... product."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
paths = sys.argv[1:] + [os.path.join(ROOT, "frequensea_amd", "libfsea_hip.so")]
vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream


def load(path):
    L = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    L.fsea_plan_create.argtypes = [ctypes.POINTER(vp), ci, ci, ci, ci]
    L.fsea_exec_u8_device.argtypes = [vp, vp, sz, ci, vp, vp]
    L.fsea_exec_u8_tiled_device.argtypes = [vp, vp, sz, ci, vp, sz, sz, sz, sz, sz, vp]
    return L


libs = [(os.path.basename(p), load(p)) for p in paths]


def ab(title, make_plan, launch, launches, rounds=9, steps_wall=20):
    plans = []
    for name, L in libs:
        p = vp()
        assert make_plan(L, p) == 0, name
        plans.append((name, L, p))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:                       # common clock pre-warm
        for name, L, p in plans:
            for k in range(8):
                launch(L, p, k)
        torch.cuda.synchronize()
    ev, wall = {n: [] for n, _, _ in plans}, {n: [] for n, _, _ in plans}
    for rnd in range(rounds):
        for name, L, p in (plans if rnd % 2 == 0 else plans[::-1]):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(launches):
                launch(L, p, k)
            e1.record()
            torch.cuda.synchronize()
            ev[name].append(e0.elapsed_time(e1) / launches)
            t1 = time.perf_counter()
            for k in range(steps_wall):
                launch(L, p, k)
            torch.cuda.synchronize()
            wall[name].append(1e3 * (time.perf_counter() - t1) / steps_wall)
    print("== " + title)
    for name, _, _ in plans:
        e, w = np.array(ev[name]), np.array(wall[name])
        print("%-24s events: median %.4f ms (min %.4f max %.4f)   wall, K=%d: median %.4f ms (min %.4f max %.4f)" %
              (name, np.median(e), e.min(), e.max(), steps_wall, np.median(w), w.min(), w.max()))


n, rows, tiles = 4096, 256, 512
gen = torch.Generator(device=dev)
gen.manual_seed(4000000)
iq = torch.clamp(torch.round(torch.randn(2 * tiles * rows * n, generator=gen, device=dev) * 20.0), -128, 127).to(torch.int8)
img = torch.empty((rows, tiles * n), dtype=torch.uint8, device=dev)
ab("config 4: 512 x 256 x 4096-pt DB5 sweep, tiles written into the stitched image",
   lambda L, p: L.fsea_plan_create(ctypes.byref(p), n, n, 2, 0),
   lambda L, p, k: L.fsea_exec_u8_tiled_device(p, iq.data_ptr(), tiles * rows, 1, img.data_ptr(), rows, tiles * n, 0, rows, n, stream),
   launches=20)
px = torch.empty(tiles * rows * n, dtype=torch.uint8, device=dev)
ab("the same sweep as a plain tile stack (fsea_exec_u8_device)",
   lambda L, p: L.fsea_plan_create(ctypes.byref(p), n, n, 2, 0),
   lambda L, p, k: L.fsea_exec_u8_device(p, iq.data_ptr(), tiles * rows, 1, px.data_ptr(), stream),
   launches=20)
del img, px
n8, f8, sets = 8192, 4096, 6
host = np.random.default_rng(1).integers(-70, 70, 2 * f8 * n8, dtype=np.int8)
ins = [torch.from_numpy(np.roll(host, 16 * s)).to(dev) for s in range(sets)]
outs = [torch.empty(f8 * n8, dtype=torch.float32, device=dev) for _ in range(sets)]
ab("headline: 8192-pt x 4096 frames MAG, six rotating buffer sets",
   lambda L, p: L.fsea_plan_create(ctypes.byref(p), n8, n8, 0, 0),
   lambda L, p, k: L.fsea_exec_u8_device(p, ins[k % sets].data_ptr(), f8, 1, outs[k % sets].data_ptr(), stream),
   launches=200)
