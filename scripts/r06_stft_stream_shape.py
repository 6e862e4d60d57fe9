#!/usr/bin/env python3
"""bench.py's multi-GPU leg reports fsea_fft16384_u8_mag_half at 0.387 of the HBM roofline for the
config-5 stream (one 32767-frame launch, timed over 5 launches straight after the checks, no clock pre-warm) while the same
kernel shows 0.44 in `extra.stft16384_*` (8191-frame launches, pre-warmed, median of 5 regions).  Which of the differences
is it?  The whole stream (32767 frames, 0.5 GiB in, 2 GiB of rows out) transformed as 1, 2, 4, 8 launches, on one stream and
alternately on two, each measured cold (5 launches from an idle chip) and warm (0.25 s pre-warm, median of 5 regions)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402

from frequensea_amd import fsea, sweep  # noqa: E402


def main():
    n, hop, frames = 16384, 8192, 32767
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    samples = (frames - 1) * hop + n
    iq = torch.clamp(torch.round(torch.randn(2 * samples, generator=gen, device=dev) * 20.0), -128, 127).to(torch.int8)
    out = torch.empty((frames, n), dtype=torch.float32, device=dev)
    plan = fsea.Plan(n, hop=hop, mode=fsea.MODE_MAG_F32, device=0)
    alg = (2 * hop + 4 * n) * frames
    streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
    res = []

    def one_pass(pieces, two):
        for j, (a, b) in enumerate(sweep.chunk_ranges(0, frames, pieces)):
            st = streams[j % 2] if two else streams[0]
            plan.exec_device(iq.data_ptr() + 2 * a * hop, b - a, out.data_ptr() + 4 * n * a, flip=True, stream=st.cuda_stream)

    for pieces, two in ((1, False), (2, False), (4, False), (8, False), (2, True), (4, True), (8, True)):
        time.sleep(0.5)                                      # idle chip
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            one_pass(pieces, two)
        torch.cuda.synchronize()
        cold = (time.perf_counter() - t0) / 5
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.25:
            one_pass(pieces, two)
            torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(5):
                one_pass(pieces, two)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 5)
        warm = float(np.median(ts))
        res.append({"launches_per_stream_pass": pieces, "two_streams": two, "cold_ms": 1e3 * cold, "warm_ms": 1e3 * warm,
                    "cold_frac": alg / cold / 1e9 / 8000.0, "warm_frac": alg / warm / 1e9 / 8000.0})
        print(json.dumps(res[-1]))
    print("checksum", "%016x" % sweep.checksum(torch, out))
    plan.close()


if __name__ == "__main__":
    main()
