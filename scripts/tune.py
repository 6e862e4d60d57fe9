#!/usr/bin/env python3
"""Times every compiled kernel variant on the GPU (HIP events on the launch stream, data set
larger than the Infinity Cache) and checks a few rows of each against numpy's FFT.
Usage: python scripts/tune.py [N ...]   -> one line per (N, variant)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frequensea_amd import fsea  # noqa: E402
fsea.use_tune_library()

VARIANTS = {
    8192: ["", "cp0", "nd", "st_nt", "ld_nt", "x0", "v2", "v2s", "A", "B", "D", "B2", "D2", "W", "W2", "notwl", "notwr",
           "abl_nostore", "abl_nolds", "abl_noflop", "abl_io", "abl_valu", "abl_noload", "abl_nomag",
           "abl_io_nt", "abl_nolds_nt", "abl_noflop_nt", "abl_v2l", "abl_v2sl", "abl_v2na", "abl_px_nolog", "abl_m16", "abl_m32"],
    1024: ["", "cp0", "ldst_nt", "x0", "B", "C", "D"],
    4096: ["", "w64", "s2", "nr", "cp0", "st_nt", "t256", "x0", "df", "B", "B3", "C", "D",
           "abl_px_nolog", "abl_px_nost", "abl_px_io"],
    32: [""], 64: [""], 128: ["", "p16"], 256: ["", "cp0", "ldst_nt", "p16"], 512: [""], 2048: ["", "nr", "cp0", "st_nt", "x0", "df", "B", "C"],
    16384: ["", "cp0", "st_nt", "nd", "B"],
}
TOTAL_SAMPLES = 1 << 27          # 256 MiB in + 512 MiB out per launch
ROUNDS = 7
# TUNE_SETS=k: k independent buffer sets used in rotation (streaming regime, nothing served from the
# Infinity Cache); TUNE_FRAMES=f: frames per launch (default: TOTAL_SAMPLES / N); TUNE_MODE=m: epilogue mode
SETS = int(os.environ.get("TUNE_SETS", "1"))
MODE = int(os.environ.get("TUNE_MODE", "0"))
OUT_BYTES = {0: 4, 1: 1, 2: 1, 3: 8, 4: 4, 5: 4}[MODE]


def dev_alloc(nbytes):
    p = ctypes.c_void_p()
    fsea._check(fsea.hip_lib().fsea_device_alloc(0, nbytes, ctypes.byref(p)))
    return p


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [8192, 1024, 4096, 16384, 2048, 512, 256, 128]
    L = fsea.hip_lib()
    rng = np.random.default_rng(1)
    host = rng.integers(-70, 70, 2 * TOTAL_SAMPLES, dtype=np.int8).view(np.uint8)
    if os.environ.get("TUNE_CONST_INPUT"):  # how much of the time is data-dependent (power)?
        host[:] = 0x80
    per_set = TOTAL_SAMPLES if not os.environ.get("TUNE_FRAMES") else int(os.environ["TUNE_FRAMES"]) * max(sizes)
    d_ins = [dev_alloc(2 * per_set) for _ in range(SETS)]
    d_outs = [dev_alloc(max(4, OUT_BYTES) * per_set) for _ in range(SETS)]
    for d in d_ins:
        fsea._check(L.fsea_copy_to_device(0, d, host.ctypes.data, 2 * per_set))
    d_in, d_out = d_ins[0], d_outs[0]
    for n in sizes:
        frames = int(os.environ["TUNE_FRAMES"]) if os.environ.get("TUNE_FRAMES") else TOTAL_SAMPLES // n
        u = (host[: 2 * n * 3] ^ np.uint8(0x80)).astype(np.float64).reshape(3, n, 2) / 256.0
        want = np.abs(np.fft.fft((u[..., 0] + 1j * u[..., 1]) * (1.0 - 2.0 * (np.arange(n) & 1)), axis=1))
        want[:, n // 2] = want[:, n // 2 - 1]                 # src/nrf.c:599-630
        plans = []
        names = VARIANTS.get(n, [""])
        if os.environ.get("TUNE_VARIANTS"):  # e.g. TUNE_VARIANTS=-,x1,x3 ("-" = the default kernel)
            names = ["" if v == "-" else v for v in os.environ["TUNE_VARIANTS"].split(",")]
        for var in names:
            try:
                plans.append((var, fsea.Plan(n, variant=var, mode=MODE)))
                if os.environ.get("TUNE_UNITS"):   # static | tickets: pin the frame distribution (default: per launch)
                    plans[-1][1].set_unit_distribution({"static": fsea.UNITS_STATIC, "tickets": fsea.UNITS_TICKETS}[os.environ["TUNE_UNITS"]])
            except fsea.FseaError as e:
                print("N=%d variant=%-6s unavailable: %s" % (n, var, e))
        # warm the clocks, then interleave the variants over several rounds so that order,
        # DVFS state and neighbours affect every variant alike; report median and best
        def timed(plan, reps):
            if SETS > 1:
                return plan.time_rotating(d_ins, frames, d_outs, reps)
            return plan.time_device(d_in, frames, d_out, reps)

        for _, plan in plans:
            timed(plan, 20)
        times = {var: [] for var, _ in plans}
        rels = {}
        for rnd in range(ROUNDS):
            order = plans if rnd % 2 == 0 else plans[::-1]
            for var, plan in order:
                times[var].append(timed(plan, 10 if SETS == 1 else 10 * SETS))
                if rnd == 0 and MODE == 0:
                    got = np.empty((3, n), np.float32)
                    fsea._check(L.fsea_copy_to_host(0, got.ctypes.data, d_out, got.nbytes))
                    rels[var] = float(np.linalg.norm(got - want) / np.linalg.norm(want))
        for var, plan in plans:
            ms = float(np.median(times[var]))
            best = float(np.min(times[var]))
            rel = rels.get(var, float("nan"))
            gbs = (2.0 + OUT_BYTES) * n * frames / (ms * 1e-3) / 1e9
            grid = plan.grid(frames)
            print("N=%-5d variant=%-11s %-28s grid=%-4d wg=%-3d lds=%-6d median %7.3f ms (best %7.3f)  %7.1f Mframes/s  "
                  "%6.1f GB/s  %4.1f%% of 8 TB/s  rel=%.1e %s"
                  % (n, var or "-", plan.kernel_name, grid[0], grid[1], grid[2], ms, best, frames / ms / 1e3, gbs,
                     gbs / 80.0, rel,
                     "OK" if rel < 1e-6 else ("ablation" if var.startswith("abl_") else ("-" if MODE else "MISMATCH"))))
            plan.close()


if __name__ == "__main__":
    main()
