#!/usr/bin/env python3
"""Soak run: minutes of randomly drawn launches (size, mode, frame count, hop, byte convention, frame distribution,
one of four streams, plain / tiled / frequency-shifted entry point), every one checked on sampled rows against numpy;
in between, the synchronous host-memory entry points (u8, frequency-shifted, f64 input: the pipelined copy-in /
transform / copy-out path) and the device-resident history ring (push, shift, get).
Looks for what the unit tests cannot: rare interleavings of overlapping launches, ticket-counter residue, hangs, and the
store-data hazard of the 16-byte row stores (a third of the launches are drawn from the sizes and modes that use them,
long enough to load every CU, with more rows sampled).
Usage: python scripts/soak.py [seconds, default 120] [seed | "random"]      (the seed is printed first)"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frequensea_amd import fsea  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
SEED = 7 if len(sys.argv) <= 2 else (int.from_bytes(os.urandom(4), "little") if sys.argv[2] == "random" else int(sys.argv[2]))
print("soak: seed %d" % SEED, flush=True)
rng = np.random.default_rng(SEED)
L = fsea.hip_lib()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipStreamCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
streams = [ctypes.c_void_p() for _ in range(4)]
for s in streams:
    assert hip.hipStreamCreate(ctypes.byref(s)) == 0
SIZES = [32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384]
MAX_IN = 1 << 27                                   # bytes of IQ per launch at most
host = rng.integers(-100, 100, MAX_IN + (1 << 16), dtype=np.int8).view(np.uint8)


def dev_alloc(nbytes):
    p = ctypes.c_void_p()
    fsea._check(L.fsea_device_alloc(0, nbytes, ctypes.byref(p)))
    return p


d_in = dev_alloc(host.nbytes)
fsea._check(L.fsea_copy_to_device(0, d_in, host.ctypes.data, host.nbytes))
d_out = [dev_alloc(4 * MAX_IN) for _ in streams]   # f32 rows of a non-overlapped launch fit; others are clipped below
plans = {}
WINDOWS = {}                                       # (name, n) -> the float32 weights handed to fsea_plan_set_window


def rows_numpy(offset_bytes, frame, n, hop, flip, mode, shift=None, window=None):
    raw = host[offset_bytes + 2 * frame * hop: offset_bytes + 2 * frame * hop + 2 * n]
    u = (raw ^ np.uint8(0x80 if flip else 0)).astype(np.float64).reshape(n, 2) / 256.0
    y = u[:, 0] + 1j * u[:, 1]
    if shift is not None:                          # fsea.h: y[m] = (u8[m]/256) e^{2 pi i (phase0 + m delta)} + 0.5 (1 + i)
        m_idx = frame * hop + np.arange(n)
        y = y * np.exp(2j * np.pi * (shift[1] + m_idx * shift[0])) + 0.5 * (1 + 1j)
    x = y * (1.0 - 2.0 * (np.arange(n) & 1))
    if window is not None:                         # the taper beside the (-1)^n (fsea_plan_set_window)
        x = x * WINDOWS[(window, n)].astype(np.float64)
    X = np.fft.fft(x)
    mag = np.abs(X)
    if mode == 0:
        mag[n // 2] = mag[n // 2 - 1]
        return mag
    if mode == 3:
        return X
    if mode == 4:
        return mag
    if mode == 5:
        return 10.0 * np.log10(mag * mag + 1e-20)
    p = mag * mag
    k = 100.0 if mode == 1 else 50.0
    px = np.clip((k * np.log10(p + 1e-20)).astype(np.int64), 0, 255)
    if mode == 2:
        px[n // 2] = px[n // 2 - 1]
    return px


def check_float(got, want, what, bound=5e-6):
    rel = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)
    assert rel < bound, (what, rel)


host_calls = 0
histories = {}


def host_entry_points():
    """One synchronous call of a host-memory entry point, or a few operations on a device-resident history ring, checked
    at once: fsea_exec_u8_host (small = mapped staging, large = the pipelined path), _shifted_host, _f64_host;
    fsea_history_push / shift / get against a numpy model of nrf_fft's history (src/nrf.c:569-631)."""
    global host_calls
    host_calls += 1
    kind = int(rng.integers(4))
    n = int(rng.choice(SIZES))
    if kind == 3:
        n = int(rng.choice([128, 1024, 4096]))
        rows = 5
        if n not in histories:
            plan = fsea.Plan(n, mode=0)
            h = ctypes.c_void_p()
            fsea._check(L.fsea_history_create(plan._p, rows, ctypes.byref(h)))
            histories[n] = (plan, h, np.zeros((rows, n)))
        plan, h, model = histories[n]
        for _ in range(int(rng.integers(1, 4))):
            off = 16 * int(rng.integers(0, 4096))
            fsea._check(L.fsea_history_push_u8_host(h, host[off:].ctypes.data, 1))
            model[1:] = model[:-1].copy()
            model[0] = rows_numpy(off, 0, n, n, True, 0)
        if rng.random() < 0.5:
            sh = int(rng.choice([-n, -7, -1, 1, 3, n // 2, n + 5]))
            fsea._check(L.fsea_history_shift(h, sh))
            new = np.zeros_like(model)
            if 0 < sh < n:
                new[:, : n - sh] = model[:, sh:]
            elif -n < sh < 0:
                new[:, -sh:] = model[:, : n + sh]
            model[:] = new
        got = np.empty((rows, n), np.float64)
        fsea._check(L.fsea_history_get_f64(h, got.ctypes.data))
        check_float(got, model, ("history", n))
        return
    mode = int(rng.choice([0, 0, 1, 3]))
    hop = n if rng.random() < 0.7 else max(8, (n // 2) - (n // 2) % 8)
    esz = {0: 4, 1: 1, 3: 8}[mode]
    nf = int(rng.choice([1, 2, 5, 33, 700, 5000]))
    nf = int(min(nf, (1 << 25) // (esz * n), ((1 << 24) - 2 * n) // (2 * hop) + 1))
    flip = bool(rng.integers(2))
    off = 16 * int(rng.integers(0, 2048))
    key = (n, hop, mode)
    if key not in plans:
        plans[key] = fsea.Plan(n, hop=hop, mode=mode)
    plan = plans[key]
    raw = host[off: off + 2 * ((nf - 1) * hop + n)]
    frames = sorted({0, nf - 1, int(rng.integers(nf))})
    if kind == 0:
        got = plan.exec_host(raw, nf, flip=flip)
        shift = None
    elif kind == 1:
        shift = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(0, 1)))
        got = plan.exec_shifted_host(raw, nf, shift[0], shift[1], flip=flip)
    else:                                          # f64 input: x[n] = (-1)^n (re + i im), no scaling, no flip
        x = rng.normal(0, 0.3, 2 * ((nf - 1) * hop + n))
        got = plan.exec_host_f64(x, nf)
        for f in frames:
            z = x[2 * f * hop: 2 * (f * hop + n)].reshape(n, 2)
            X = np.fft.fft((z[:, 0] + 1j * z[:, 1]) * (1.0 - 2.0 * (np.arange(n) & 1)))
            if mode == 0:
                want = np.abs(X)
                want[n // 2] = want[n // 2 - 1]
            elif mode == 3:
                want = X
            else:
                continue                           # (pixel rows of f64 input: covered by the unit tests)
            check_float(got[f], want, ("f64 host", n, nf, hop, mode, f))
        return
    for f in frames:
        want = rows_numpy(off, f, n, hop, flip, mode, shift)
        if mode == 1:
            delta = np.abs(got[f].astype(np.int32) - want.astype(np.int32))
            assert delta.max() <= 1 and (delta != 0).sum() <= max(4, n // 50), ("host", kind, n, nf, hop, flip, f)
        else:
            check_float(got[f], want, ("host", kind, n, nf, hop, flip, mode, f))


t_end = time.time() + SECONDS
launches = checked = 0
pending = [None] * len(streams)
while time.time() < t_end:
    si = int(rng.integers(len(streams)))
    if pending[si] is not None:                    # verify what this stream ran last, then reuse its buffer
        assert hip.hipStreamSynchronize(streams[si]) == 0
        n, nf, hop, flip, mode, off, tiled, shape, shift, wname = pending[si]
        dt = {0: np.float32, 1: np.uint8, 2: np.uint8, 3: np.complex64, 4: np.float32, 5: np.float32}[mode]
        # sizes / modes whose rows leave in 16-byte stores (four adjacent f32 bins or two complex bins per lane): sample more
        wide = (mode in (0, 4, 5) and n in (64, 128, 2048)) or (mode == 3 and n in (32, 64, 128, 256, 512, 2048, 4096))
        extra = [int(x) for x in rng.integers(0, nf, 12)] if wide else []
        for f in sorted({0, nf - 1, int(rng.integers(nf))} | set(extra)):
            row = np.empty(n, dt)
            if tiled:
                rows_t, stride, first_x, step = shape
                k, y = divmod(f, rows_t)
                src = d_out[si].value + (y * stride + first_x + k * step) * row.itemsize
            else:
                src = d_out[si].value + f * n * row.itemsize
            fsea._check(L.fsea_copy_to_host(0, row.ctypes.data, ctypes.c_void_p(src), row.nbytes))
            want = rows_numpy(off, f, n, hop, flip, mode, shift, wname)
            if shift is not None and mode in (0, 2):   # the restored offset sits in bin N/2, which these modes overwrite
                pass
            if mode in (1, 2):
                # f32 transform against an f64 reference: a pixel on a truncation boundary may land one level off
                # (SURVEY 8(c)); on a single row that is a handful of pixels, never more than one level
                delta = np.abs(row.astype(np.int32) - want.astype(np.int32))
                assert delta.max() <= 1 and (delta != 0).sum() <= max(4, n // 50), (n, nf, hop, flip, mode, f, int(delta.max()), int((delta != 0).sum()))
            else:
                if mode == 5:                      # dB rows: back to magnitudes, then the same relative-L2 bound
                    got_mag, want_mag = 10.0 ** (row.astype(np.float64) / 20.0), 10.0 ** (want / 20.0)
                    rel = np.linalg.norm(got_mag - want_mag) / max(np.linalg.norm(want_mag), 1e-30)
                    assert rel < 1e-5, (n, nf, hop, flip, mode, f, rel)
                else:
                    rel = np.linalg.norm(row - want) / max(np.linalg.norm(want), 1e-30)
                    # (a faulty row is off by tens of per cent; the bounds only have to sit above fp32 round-off: 1-2e-7 typical)
                    assert rel < 5e-6, (n, nf, hop, flip, mode, f, rel)
            checked += 1
        pending[si] = None
    if rng.random() < 0.08:
        host_entry_points()
        continue
    n = int(rng.choice(SIZES))
    mode = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5]))
    hazard_draw = rng.random() < 0.33
    if hazard_draw:                                # the 16-byte-store kernels, long launches
        n = int(rng.choice([64, 128, 1024, 2048]))
        mode = int(rng.choice([0, 0, 3, 4, 5]))
    anysize = (not hazard_draw) and rng.random() < 0.04
    if anysize:                                    # sizes without a kernel of their own: Bluestein / four-step (plan-owned work buffers)
        n = int(rng.choice([48, 1000, 1023, 6000, 32768]))
    hop = n if (hazard_draw or anysize or rng.random() < 0.7) else int(rng.choice([n // 2, n // 4, 2 * n, 8]))
    hop = max(8, hop - hop % 8) if not anysize else n
    esz = {0: 4, 1: 1, 2: 1, 3: 8, 4: 4, 5: 4}[mode]
    nf_max = min((MAX_IN - 2 * n) // (2 * hop) + 1, (4 * MAX_IN) // (esz * n))
    nf = int(min(nf_max, rng.choice([1, 2, 3, 7, 64, 511, 4096, 20000, 70000])))
    if anysize:
        nf = min(nf, 300)
    if hazard_draw:
        nf = int(min(nf_max, max(nf, (1 << 23) // n)))
    flip = bool(rng.integers(2))
    off = 16 * int(rng.integers(0, 2048))
    # a plan with a taper window (fused into pass 0; centred form for "hann", offset-binary form for "noise") now and then
    wname = None if (anysize or rng.random() >= 0.2) else str(rng.choice(["hann", "noise"]))
    key = (n, hop, mode, wname)
    if key not in plans:
        plans[key] = fsea.Plan(n, hop=hop, mode=mode)
        if wname:
            if (wname, n) not in WINDOWS:
                WINDOWS[(wname, n)] = fsea.window("hann", n) if wname == "hann" else np.random.default_rng(n).uniform(-1, 2, n).astype(np.float32)
            plans[key].set_window(WINDOWS[(wname, n)])
    plan = plans[key]
    plan.set_unit_distribution(int(rng.integers(3)))
    tiled, shape, shift = False, None, None
    if rng.random() < 0.12 and not anysize:        # the frequency shifter fused into the load (fsea_exec_u8_shifted_device); with a taper: *_u8_rot_win
        shift = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(0, 1)))
        plan.exec_shifted_device(ctypes.c_void_p(d_in.value + off), nf, d_out[si], shift[0], shift[1], flip=flip,
                                 stream=streams[si].value)
    elif hop == n and not anysize and rng.random() < 0.2:            # rows into an image, tiles side by side
        fpw = 1 if n >= 8192 else 2 if n == 4096 else 4 if n == 2048 else 8 if n == 1024 else 16 if n == 512 else 64
        rows_t = fpw * int(rng.integers(1, 4))
        tiles = max(1, min(nf // rows_t, 6))
        nf = tiles * rows_t
        step, first_x = n + 4 * int(rng.integers(0, 3)), 4 * int(rng.integers(0, 5))
        stride = first_x + (tiles - 1) * step + n + 8
        tiled, shape = True, (rows_t, stride, first_x, step)
        plan.exec_tiled_device(ctypes.c_void_p(d_in.value + off), nf, d_out[si], rows_t + 1, stride, first_x, rows_t, step,
                               flip=flip, stream=streams[si].value)
    else:
        plan.exec_device(ctypes.c_void_p(d_in.value + off), nf, d_out[si], flip=flip, stream=streams[si].value)
    pending[si] = (n, nf, hop, flip, mode, off, tiled, shape, shift, wname)
    launches += 1
for s in streams:
    assert hip.hipStreamSynchronize(s) == 0
print("soak: seed %d, %.0f s, %d launches on %d streams, %d host-memory / history calls, %d plans, %d rows checked against numpy, "
      "no mismatch, no hang" % (SEED, SECONDS, launches, len(streams), host_calls, len(plans), checked))
