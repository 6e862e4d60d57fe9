#!/usr/bin/env python3
"""Device parity of named kernel variants (tuning library) against the oracle: every u8 epilogue mode, the
compile-time and the run-time-mode kernels, long launches (every CU loaded) and identical-launch determinism.
Usage: [CHECK_MODES=0,1,2] python scripts/check_variant.py N variant [variant ...]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frequensea_amd import fsea  # noqa: E402
fsea.use_tune_library()
from tests import parity  # noqa: E402


def main():
    n = int(sys.argv[1])
    variants = ["" if v == "-" else v for v in sys.argv[2:]] or [""]
    L = fsea.hip_lib()
    rng = np.random.default_rng(7)
    frames = max(4096, (1 << 25) // n)            # long enough to put several units on every workgroup
    host = rng.integers(-90, 90, 2 * n * frames, dtype=np.int8).view(np.uint8)
    d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    fsea._check(L.fsea_device_alloc(0, host.nbytes, ctypes.byref(d_in)))
    fsea._check(L.fsea_device_alloc(0, 8 * n * frames, ctypes.byref(d_out)))
    fsea._check(L.fsea_copy_to_device(0, d_in, host.ctypes.data, host.nbytes))
    check = sorted(set(list(range(6)) + list(rng.integers(0, frames, 26)) + [frames - 1]))
    bad = 0
    for var in variants:
        for mode in [int(m) for m in os.environ.get("CHECK_MODES", "0,1,2,3,4,5").split(",")]:
            for flip in (True, False):
                plan = fsea.Plan(n, mode=mode, variant=var)
                outs = []
                for rep in range(3):
                    fsea._check(L.fsea_copy_to_device(0, d_out, np.zeros(16, np.uint8).ctypes.data, 16))
                    plan.exec_device(d_in, frames, d_out, flip=flip)
                    plan.synchronize()
                    got = np.empty((frames, n), dtype=plan.out_dtype)
                    fsea._check(L.fsea_copy_to_host(0, got.ctypes.data, d_out, got.nbytes))
                    outs.append(got)
                same = all(np.array_equal(outs[0], o) for o in outs[1:])
                try:
                    for f in check:
                        parity.check_mode(outs[0][f:f + 1], host[2 * n * f: 2 * n * (f + 1)], n, 1, n, flip, mode)
                    ok = "OK"
                except AssertionError as e:
                    ok = "MISMATCH " + str(e)[:160]
                    bad += 1
                if not same:
                    bad += 1
                print("N=%d variant=%-6s mode=%d flip=%d %-26s %s  identical-launches=%s" %
                      (n, var or "-", mode, flip, plan.kernel_name, ok, same))
                plan.close()
    print("check_variant:", "ALL OK" if bad == 0 else "%d FAILURES" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
