#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's recorded IQ captures.

Run in the build container only (needs /root/reference/rfdata).  The expected
outputs are computed here with numpy + scipy.fft (pocketfft, double precision)
as an implementation that is independent of oracle/fsea_oracle.c; the tests
then pin the C oracle (and, on the GPU, the HIP path) against these files.

What each expected array restates (paths relative to /root/reference):
  flipped   src/nrf.c:100-109       u = (b + 128) % 256
  mag       src/nrf.c:601-630       |FFT((-1)^n * u/256)|, bin N/2 := bin N/2-1
  spec      src/nrf.c:615           the complex spectrum itself (FFTW forward)
  db10      c/fft-batch.c:83-94     clamp_u8(trunc(10*log10(p+1e-20)*10))
  db5       c/fft-batch-broad.c:106-121  ... *5, pixel N/2 := pixel N/2-1
  shift_*   src/nrf.c:569-596       nrf_fft_shift on a 4-row history
  hann_mag  the same with a periodic Hann taper beside the (-1)^n (BASELINE.json's windowed FFT; an
            extension, the reference has no taper): scipy.signal.get_window("hann", N) rounded to f32 -- the
            weights fsea_window_fill hands the kernels
The inputs are the first 2*16384 bytes of four 262144-byte captures
(data files, not source); their sha256 prefixes are recorded in the npz.

rfdata_all_golden.npz (round 4): EVERY full-size capture of the reference's rfdata/ (35 files; rf-433.000-short.raw
is 172144 bytes, for which the reference's replay computes zero blocks) -- first 2*8192 bytes,
sha256 prefix, magnitude rows at N = 1024 and 8192 -- plus, for rf-202.500-1.raw, the whole 262144-byte block and
the 128 rows of its 128 consecutive 1024-point frames.
"""
import hashlib
import os
import sys

import numpy as np
import scipy.fft

REF = "/root/reference/rfdata"
FILES = ["rf-100.900-1.raw", "rf-202.500-1.raw", "rf-202.500-2.raw", "rf-202.500-3.raw"]
SIZES = [128, 256, 1024, 4096, 8192, 16384]
HEAD_BYTES = 2 * 16384
HERE = os.path.dirname(os.path.abspath(__file__))


def spectrum(raw_u8, n, window=None):
    u = ((raw_u8[: 2 * n].astype(np.int32) + 128) % 256).astype(np.float64) / 256.0
    x = u[0::2] + 1j * u[1::2]
    x = x * np.where(np.arange(n) % 2 == 0, 1.0, -1.0)
    if window is not None:
        x = x * window
    return scipy.fft.fft(x)


def mag_row(spec):
    n = spec.size
    row = np.sqrt(spec.real * spec.real + spec.imag * spec.imag)
    row[n // 2] = row[n // 2 - 1]
    return row


def db_row(spec, scale, dcfix):
    pwr = spec.real * spec.real + spec.imag * spec.imag
    v = 10.0 * np.log10(pwr + 1.0e-20) * scale
    v = np.clip(np.trunc(v), 0, 255).astype(np.uint8)
    if dcfix:
        v[spec.size // 2] = v[spec.size // 2 - 1]
    return v


def shift_ref(hist, d):
    n = hist.shape[1]
    s = int(np.floor(abs(n / d) + 0.5)) * (1 if n / d >= 0 else -1)  # C round(): half away from zero
    out = hist.copy()
    if s == 0:
        return out
    if abs(s) >= n:
        out[:] = 0
        return out
    if s > 0:
        out[:, : n - s] = hist[:, s:]
        out[:, n - s:] = 0
    else:
        out[:, -s:] = hist[:, : n + s]
        out[:, :-s] = 0
    return out


def capture_key(fname):
    return fname.replace(".raw", "").replace("-", "_").replace(".", "p")


def all_captures():
    """rfdata_all_golden.npz: every full-size capture, rows at 1024 and 8192; one capture's whole block at 1024."""
    store = {}
    names = sorted(f for f in os.listdir(REF) if f.endswith(".raw") and os.path.getsize(os.path.join(REF, f)) == 262144)
    for fname in names:
        raw = np.fromfile(os.path.join(REF, fname), dtype=np.uint8)
        key = capture_key(fname)
        store[key + "__sha256"] = np.frombuffer(hashlib.sha256(raw.tobytes()).hexdigest()[:16].encode(), dtype=np.uint8)
        store[key + "__raw"] = raw[:2 * 8192].copy()
        for n in (1024, 8192):
            store["%s__mag_%d" % (key, n)] = mag_row(spectrum(raw, n))
    raw = np.fromfile(os.path.join(REF, "rf-202.500-1.raw"), dtype=np.uint8)
    store["block__name"] = np.frombuffer(b"rf_202p500_1", dtype=np.uint8)
    store["block__raw"] = raw
    store["block__mag_1024"] = np.stack([mag_row(spectrum(raw[2 * 1024 * f:], 1024)) for f in range(128)])
    out = os.path.join(HERE, "rfdata_all_golden.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, os.path.getsize(out), "bytes,", len(store), "arrays,", len(names), "captures")


def main():
    if not os.path.isdir(REF):
        print("reference rfdata not present; nothing generated", file=sys.stderr)
        return 1
    all_captures()
    import scipy.signal
    store = {}
    for fname in FILES:
        raw = np.fromfile(os.path.join(REF, fname), dtype=np.uint8)
        assert raw.size == 262144
        key = fname.replace(".raw", "").replace("-", "_").replace(".", "p")
        store[key + "__sha256"] = np.frombuffer(
            hashlib.sha256(raw.tobytes()).hexdigest()[:16].encode(), dtype=np.uint8)
        head = raw[:HEAD_BYTES].copy()
        store[key + "__raw"] = head
        store[key + "__flipped"] = ((head.astype(np.int32) + 128) % 256).astype(np.uint8)
        for n in SIZES:
            spec = spectrum(raw, n)
            store["%s__mag_%d" % (key, n)] = mag_row(spec)
            store["%s__db10_%d" % (key, n)] = db_row(spec, 10.0, False)
            store["%s__db5_%d" % (key, n)] = db_row(spec, 5.0, True)
            if n <= 1024:
                store["%s__spec_%d" % (key, n)] = spec
            if n in (1024, 8192, 16384):
                w = scipy.signal.get_window("hann", n).astype(np.float32).astype(np.float64)
                store["%s__hann_mag_%d" % (key, n)] = mag_row(spectrum(raw, n, w))
    # nrf_fft_shift on a 4-row, 256-bin history made of the four captures
    hist = np.stack([store[k.replace(".raw", "").replace("-", "_").replace(".", "p") + "__mag_256"]
                     for k in FILES])
    store["shift__history"] = hist
    for name, d in [("p8", 8.0), ("m8", -8.0), ("half", 0.5), ("p50", 50.0),
                    ("m3", -3.0), ("big", 1.0e9), ("mhalf", -0.5)]:
        store["shift__" + name] = shift_ref(hist, d)
        store["shift__" + name + "__d"] = np.array([d])
    out = os.path.join(HERE, "rfdata_golden.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, os.path.getsize(out), "bytes,", len(store), "arrays")
    return 0


if __name__ == "__main__":
    sys.exit(main())
