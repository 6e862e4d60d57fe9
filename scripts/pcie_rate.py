#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline batch through the host-buffer entry point fsea_exec_u8_host (never bench.py's
`value`): 4096 frames of 8192 points, 64 MiB in + 128 MiB out, three kinds of caller memory."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frequensea_amd import fsea

n, frames = 8192, 4096
iq = np.random.default_rng(0).integers(-70, 70, 2 * n * frames, dtype=np.int8).view(np.uint8)
plan = fsea.Plan(n)
ref = plan.exec_host(iq, frames)


def report(what, dt):
    print("fsea_exec_u8_host N=%d x %d frames, %-58s %6.2f ms -> %.2f Mframes/s, %.1f Gsamples/s, %.1f GB/s over PCIe"
          % (n, frames, what + ":", dt * 1e3, frames / dt / 1e6, frames * n / dt / 1e9, 6.0 * n * frames / dt / 1e9))


def timed(fn, reps=7):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


# (a) as round 1 measured it: a fresh, never-touched output array per call (its page faults are inside the call)
report("pageable in, FRESH pageable out (np.empty per call)", timed(lambda: plan.exec_host(iq, frames)))
# (b) caller-owned pageable buffers that exist across calls (what a C caller's malloc'ed buffers are)
out = np.zeros((frames, n), np.float32)
report("pageable in, reused pageable out", timed(lambda: plan.exec_host_into(iq, frames, out)))
assert np.array_equal(out, ref)
# (c) buffers from fsea_host_alloc (pinned)
pin_in, pin_out = fsea.PinnedArray(iq.shape, np.uint8), fsea.PinnedArray((frames, n), np.float32)
pin_in.array[:] = iq
report("pinned in / out (fsea_host_alloc)", timed(lambda: plan.exec_host_into(pin_in.array, frames, pin_out.array)))
assert np.array_equal(pin_out.array, ref)
# the u8 pixel mode returns 1 byte per sample: 64 MiB in + 32 MiB out
plan_px = fsea.Plan(n, mode=fsea.MODE_DB10_U8)
px = np.zeros((frames, n), np.uint8)
dt = timed(lambda: plan_px.exec_host_into(iq, frames, px))
print("fsea_exec_u8_host N=%d x %d frames, DB10_U8 pixels (64 MiB in, 32 MiB out), reused pageable buffers: %6.2f ms -> %.2f Mframes/s"
      % (n, frames, dt * 1e3, frames / dt / 1e6))
pin_in.close(); pin_out.close()
