#!/usr/bin/env python3
"""BASELINE config 2: the nrf_* API as lua/fft.lua drives it -- nrf_fft_new(1024, 1024), then per
rendered frame nrf_fft_process(samples_buffer) + nrf_fft_get_buffer().  Reports per-call times for the
host history ring (default) and the device-resident ring (NRF_FFT_HISTORY=device)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frequensea_amd import nrf

L = nrf.nrf_lib()
rng = np.random.default_rng(0)
block = rng.integers(0, 256, nrf.NRF_BUFFER_SIZE_BYTES, dtype=np.uint8)      # one device block, offset binary
buf = L.nut_buffer_new_u8(nrf.NRF_SAMPLES_LENGTH, 2, block.ctypes.data)
for mode in ("host", "device"):
    os.environ["NRF_FFT_HISTORY"] = mode
    for n, h in ((1024, 1024), (128, 512)):
        fft = L.nrf_fft_new(n, h)
        for _ in range(20):
            L.nrf_fft_process(fft, buf)
        t0 = time.perf_counter()
        reps = 2000
        for _ in range(reps):
            L.nrf_fft_process(fft, buf)
        t1 = time.perf_counter()
        g = 200
        for _ in range(g):
            out = L.nrf_fft_get_buffer(fft)
            L.nut_buffer_free(out)
        t2 = time.perf_counter()
        s = 50
        for k in range(s):
            L.nrf_fft_shift(fft, 8.0 if k % 2 == 0 else -8.0)
        t3 = time.perf_counter()
        # the scenes' pattern: one process + one get_buffer per rendered frame
        for _ in range(g):
            L.nrf_fft_process(fft, buf)
            out = L.nrf_fft_get_buffer(fft)
            L.nut_buffer_free(out)
        t4 = time.perf_counter()
        print("history=%-6s nrf_fft(%d,%d): process %.1f us/call (%.0f rows/s), get_buffer %.1f us/call, shift %.1f us/call, "
              "process+get_buffer %.1f us per rendered frame"
              % (mode, n, h, (t1 - t0) / reps * 1e6, reps / (t1 - t0), (t2 - t1) / g * 1e6, (t3 - t2) / s * 1e6,
                 (t4 - t3) / g * 1e6))
        L.nrf_fft_free(fft)
L.nut_buffer_free(buf)
