#!/usr/bin/env python3
"""Is the gain of two streams only the ramp and the tail of a launch?  The same 2^19 frames of 8192 points, streamed
through the same 6 GiB of buffers, issued as launches of 4096 / 16384 / 65536 frames on one stream and alternately on two.
If two streams still win at 65536 frames per launch (ramp + tail < 1 % there), something other than the launch edges is
overlapped: workgroups of two launches on one CU are out of phase with each other, those of one launch are not.
Usage: python scripts/two_stream_lengths.py [N]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from frequensea_amd import fsea  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
TOTAL = (1 << 32) // N               # frames per pass: 4 Gi samples
POOL = (1 << 30) // N * 2            # frames of buffer: 2 Gi samples = 4 GiB in + 8 GiB out


def main():
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    host = rng.integers(-70, 70, 2 * N * 4096, dtype=np.int8).view(np.uint8)
    d_in = torch.from_numpy(host).to(dev).repeat(POOL // 4096)
    d_out = torch.empty(POOL * N, dtype=torch.float32, device=dev)
    plan = fsea.Plan(N, mode=fsea.MODE_MAG_F32, device=0)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def run(frames, two):
        launches = TOTAL // frames
        slots = POOL // frames
        for s in streams:
            s.synchronize()
        t0 = time.perf_counter()
        for k in range(launches):
            off = (k % slots) * frames
            plan.exec_device(d_in.data_ptr() + 2 * N * off, frames, d_out.data_ptr() + 4 * N * off, flip=True,
                             stream=streams[k % 2 if two else 0].cuda_stream)
        for s in streams:
            s.synchronize()
        return TOTAL / (time.perf_counter() - t0)

    lengths = [4096, 16384, 65536]
    for f in lengths:            # warm
        run(f, False), run(f, True)
    res = {(f, two): [] for f in lengths for two in (False, True)}
    for rnd in range(7):
        for f in lengths:
            for two in ((False, True) if rnd % 2 == 0 else (True, False)):
                res[(f, two)].append(run(f, two))
    print("N = %d, %d frames per pass through %d frames of buffers (streaming), kernel %s" % (N, TOTAL, POOL, plan.kernel_name))
    for f in lengths:
        a, b = float(np.median(res[(f, False)])), float(np.median(res[(f, True)]))
        print("frames/launch %6d: one stream %7.2f M frames/s (%.3f of 8 TB/s), two streams %7.2f M (%.3f)  x%.3f"
              % (f, a / 1e6, a * 6 * N / 8e12, b / 1e6, b * 6 * N / 8e12, b / a))
    plan.close()


if __name__ == "__main__":
    main()
