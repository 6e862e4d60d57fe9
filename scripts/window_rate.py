#!/usr/bin/env python3
"""What the fused taper window costs: the un-windowed kernel against the windowed one (product configuration, and the
two places the weights can live: tuning variants "w1" = fetched per frame, "w2" = register-resident), interleaved in one
process, streaming regime (rotating buffer sets larger than the Infinity Cache), Hann weights; also hop = N/2 at 8192
and 16384 (the half-overlap kernels, BASELINE.json config 5).  Checks three rows of every windowed kernel against numpy.
Usage: python scripts/window_rate.py [N ...]   (WINDOW_SETS=k buffer sets, default 4; WINDOW_MODE=m epilogue mode)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frequensea_amd import fsea  # noqa: E402
fsea.use_tune_library()

TOTAL_SAMPLES = 1 << 27          # 256 MiB in + 512 MiB out per set
SETS = int(os.environ.get("WINDOW_SETS", "3"))
MODE = int(os.environ.get("WINDOW_MODE", "0"))
OUT_BYTES = {0: 4, 1: 1, 2: 1, 3: 8, 4: 4, 5: 4}[MODE]
ROUNDS = 11


def dev_alloc(nbytes):
    p = ctypes.c_void_p()
    fsea._check(fsea.hip_lib().fsea_device_alloc(0, nbytes, ctypes.byref(p)))
    return p


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [256, 1024, 2048, 4096, 8192, 16384]
    L = fsea.hip_lib()
    rng = np.random.default_rng(1)
    host = rng.integers(-70, 70, 2 * TOTAL_SAMPLES, dtype=np.int8).view(np.uint8)
    d_ins = [dev_alloc(host.nbytes) for _ in range(SETS)]
    d_outs = [dev_alloc(max(4, OUT_BYTES) * TOTAL_SAMPLES) for _ in range(SETS)]
    for d in d_ins:
        fsea._check(L.fsea_copy_to_device(0, d, host.ctypes.data, host.nbytes))
    for n in sizes:
        for hop in ([n] + ([n // 2] if n >= 8192 else [])):
            frames = TOTAL_SAMPLES // n - (1 if hop < n else 0)   # rows fill the output buffer either way
            w = fsea.window("hann", n)
            u = (host[: 2 * (2 * hop + n)] ^ np.uint8(0x80)).astype(np.float64).reshape(-1, 2) / 256.0
            x = (u[:, 0] + 1j * u[:, 1])
            rows = np.stack([x[f * hop: f * hop + n] for f in range(3)]) * (1.0 - 2.0 * (np.arange(n) & 1))
            want = np.abs(np.fft.fft(rows * w.astype(np.float64), axis=1))
            want[:, n // 2] = want[:, n // 2 - 1]
            plans = [("rect", fsea.Plan(n, hop=hop, mode=MODE))]
            for var in (None, "w1", "w2"):     # None = the product plan (its mode's configuration of the size)
                try:
                    p = fsea.Plan(n, hop=hop, mode=MODE, variant=var)
                except fsea.FseaError:
                    continue
                p.set_window(w)
                plans.append(("win" + (":" + var if var else ""), p))
            for _, plan in plans:
                plan.time_rotating(d_ins, frames, d_outs, 4 * SETS)
            times = {k: [] for k, _ in plans}
            rels = {}
            for rnd in range(ROUNDS):
                for k, plan in (plans if rnd % 2 == 0 else plans[::-1]):
                    times[k].append(plan.time_rotating(d_ins, frames, d_outs, 5 * SETS))
                    if rnd == 0 and MODE == 0 and k != "rect":
                        got = np.empty((3, n), np.float32)
                        fsea._check(L.fsea_copy_to_host(0, got.ctypes.data, d_outs[(5 * SETS - 1) % SETS], got.nbytes))
                        rels[k] = float(np.linalg.norm(got - want) / np.linalg.norm(want))
            base = float(np.median(times["rect"]))
            for k, plan in plans:
                ms = float(np.median(times[k]))
                gbs = (2.0 * hop + OUT_BYTES * n) * frames / (ms * 1e-3) / 1e9
                print("N=%-5d hop=%-5d %-7s %-30s median %7.4f ms (best %7.4f) %8.2f Mframes/s %7.1f GB/s %5.1f%% of 8 TB/s  "
                      "x%.3f of rect%s" % (n, hop, k, plan.kernel_name, ms, float(np.min(times[k])), frames / ms / 1e3, gbs,
                                          gbs / 80.0, base / ms,
                                          "" if k not in rels else ("  rel=%.1e %s" % (rels[k], "OK" if rels[k] < 1e-6 else "MISMATCH"))))
                plan.close()


if __name__ == "__main__":
    main()
