"""ctypes view of the CPU oracle (oracle/libfsea_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product never imports this module.
Parity status: see oracle/fsea_oracle.h ("parity unpinned by the reference").
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MODE_MAG = 0
MODE_DB10_U8 = 1
MODE_DB5_U8_DCFIX = 2
MODE_COMPLEX = 3
MODE_MAG_NODC = 4
MODE_DB_F64 = 5


def build():
    """Compile the oracle (and oracle/_ref when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libfsea_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        c_u8p = ctypes.POINTER(ctypes.c_uint8)
        c_f64p = ctypes.POINTER(ctypes.c_double)
        L.orc_flip_u8.argtypes = [c_u8p, c_u8p, ctypes.c_size_t]
        L.orc_flip_u8.restype = None
        L.orc_unpack_center_u8.argtypes = [c_u8p, ctypes.c_size_t, c_f64p]
        L.orc_unpack_center_u8.restype = None
        L.orc_unpack_center_f64.argtypes = [c_f64p, ctypes.c_size_t, c_f64p]
        L.orc_unpack_center_f64.restype = None
        L.orc_fft_forward.argtypes = [c_f64p, c_f64p, ctypes.c_int]
        L.orc_fft_forward.restype = ctypes.c_int
        L.orc_dft_naive.argtypes = [c_f64p, c_f64p, ctypes.c_int]
        L.orc_dft_naive.restype = None
        L.orc_mag_row.argtypes = [c_f64p, ctypes.c_int, c_f64p]
        L.orc_mag_row.restype = None
        L.orc_history_scroll.argtypes = [c_f64p, ctypes.c_int, ctypes.c_int]
        L.orc_history_scroll.restype = None
        L.orc_fft_shift.argtypes = [c_f64p, ctypes.c_int, ctypes.c_int, ctypes.c_double]
        L.orc_fft_shift.restype = None
        L.orc_db_u8_row.argtypes = [c_f64p, ctypes.c_int, ctypes.c_double, ctypes.c_int, c_u8p]
        L.orc_db_u8_row.restype = None
        L.orc_mean_magnitude.argtypes = [c_f64p, ctypes.c_size_t]
        L.orc_mean_magnitude.restype = ctypes.c_double
        L.orc_composite_max.argtypes = [c_u8p, c_u8p] + [ctypes.c_uint32] * 8
        L.orc_composite_max.restype = None
        L.orc_rows.argtypes = [c_u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t,
                               ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.orc_rows.restype = ctypes.c_int
        L.orc_freq_shift.argtypes = [c_u8p, c_f64p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, c_f64p, c_f64p, c_f64p]
        L.orc_freq_shift.restype = None
        L.orc_rows_shifted.argtypes = [c_u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
        L.orc_rows_shifted.restype = ctypes.c_int
        L.orc_time_mag_rows.argtypes = [c_u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t, c_f64p]
        L.orc_time_mag_rows.restype = ctypes.c_double
        L.orc_time_mag_rows_mt.argtypes = [c_u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t,
                                           ctypes.c_int, c_f64p]
        L.orc_time_mag_rows_mt.restype = ctypes.c_double
        L.orc_time_mag_rows_fftw.argtypes = [ctypes.c_char_p, c_u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t,
                                             ctypes.c_int, ctypes.c_int, c_f64p, ctypes.c_size_t, c_f64p]
        L.orc_time_mag_rows_fftw.restype = ctypes.c_double
        _LIB = L
    return _LIB


# FFTW3-API libraries to try for the CPU baseline, in order: FFTW itself, then Intel MKL, whose
# libmkl_rt exports the FFTW3 interface (fftw_plan_dft_1d / fftw_execute) on top of its own DFT.
FFTW_CANDIDATES = ["libfftw3.so.3", "libfftw3.so", "/opt/conda/lib/libmkl_rt.so", "libmkl_rt.so", "libmkl_rt.so.1"]


def find_fftw_api():
    """First loadable FFTW3-API library (path/soname) or None."""
    os.environ.setdefault("MKL_NUM_THREADS", "1")      # one plan per thread, no nested threading
    for cand in FFTW_CANDIDATES:
        try:
            h = ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            continue
        if hasattr(h, "fftw_plan_dft_1d") and hasattr(h, "fftw_execute"):
            return cand
    return None


def time_mag_rows_fftw(lib_name, iq, n_frames, n, hop=None, threads=1, keep_rows=0, passes=1):
    """Reference-shaped loop with an FFTW3-API library doing the transform: (seconds for `passes`
    walks over the frames, rows or None)."""
    hop = n if hop is None else hop
    iq = np.ascontiguousarray(iq, dtype=np.uint8).ravel()
    rows_ = np.zeros((keep_rows, n), np.float64) if keep_rows else None
    chk = ctypes.c_double(0)
    t = lib().orc_time_mag_rows_fftw(lib_name.encode(), _u8p(iq), n_frames, n, hop, threads, passes,
                                     _f64p(rows_) if keep_rows else None, keep_rows, ctypes.byref(chk))
    if t < 0:
        raise RuntimeError("orc_time_mag_rows_fftw(%s) failed: %g" % (lib_name, t))
    return t, rows_


def _u8p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _f64p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def flip_u8(raw):
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    out = np.empty_like(raw)
    lib().orc_flip_u8(_u8p(raw), _u8p(out), raw.size)
    return out


def unpack_center_u8(iq_u8):
    iq_u8 = np.ascontiguousarray(iq_u8, dtype=np.uint8)
    n = iq_u8.size // 2
    out = np.empty(2 * n, dtype=np.float64)
    lib().orc_unpack_center_u8(_u8p(iq_u8), n, _f64p(out))
    return out.view(np.complex128)


def unpack_center_f64(iq_f64):
    iq_f64 = np.ascontiguousarray(iq_f64, dtype=np.float64)
    n = iq_f64.size // 2
    out = np.empty(2 * n, dtype=np.float64)
    lib().orc_unpack_center_f64(_f64p(iq_f64), n, _f64p(out))
    return out.view(np.complex128)


def fft_forward(x):
    x = np.ascontiguousarray(x, dtype=np.complex128)
    out = np.empty_like(x)
    rc = lib().orc_fft_forward(_f64p(x.view(np.float64)), _f64p(out.view(np.float64)), x.size)
    if rc != 0:
        raise ValueError("orc_fft_forward: n must be a power of two")
    return out


def dft_naive(x):
    x = np.ascontiguousarray(x, dtype=np.complex128)
    out = np.empty_like(x)
    lib().orc_dft_naive(_f64p(x.view(np.float64)), _f64p(out.view(np.float64)), x.size)
    return out


def mag_row(spectrum):
    s = np.ascontiguousarray(spectrum, dtype=np.complex128)
    out = np.empty(s.size, dtype=np.float64)
    lib().orc_mag_row(_f64p(s.view(np.float64)), s.size, _f64p(out))
    return out


def history_scroll(history, n, h):
    assert history.dtype == np.float64 and history.flags.c_contiguous
    lib().orc_history_scroll(_f64p(history), n, h)


def fft_shift(history, n, h, d):
    assert history.dtype == np.float64 and history.flags.c_contiguous
    lib().orc_fft_shift(_f64p(history), n, h, float(d))


def db_u8_row(spectrum, scale, dcfix):
    s = np.ascontiguousarray(spectrum, dtype=np.complex128)
    out = np.empty(s.size, dtype=np.uint8)
    lib().orc_db_u8_row(_f64p(s.view(np.float64)), s.size, float(scale), int(dcfix), _u8p(out))
    return out


def mean_magnitude(spectrum):
    s = np.ascontiguousarray(spectrum, dtype=np.complex128).ravel()
    return lib().orc_mean_magnitude(_f64p(s.view(np.float64)), s.size)


def composite_max(dst, src, dst_x, dst_y=0):
    assert dst.dtype == np.uint8 and src.dtype == np.uint8
    assert dst.flags.c_contiguous and src.flags.c_contiguous
    h, w = src.shape
    lib().orc_composite_max(_u8p(dst), _u8p(src), dst_x, dst_y, 0, 0, w, h, dst.shape[1], w)


def rows(iq, n_frames, n, hop=None, flip=True, mode=MODE_MAG):
    """Whole rows: frame f = samples [f*hop, f*hop+n) of the u8 IQ stream."""
    hop = n if hop is None else hop
    iq = np.ascontiguousarray(iq, dtype=np.uint8).ravel()
    need = 2 * ((n_frames - 1) * hop + n) if n_frames else 0
    if iq.size < need:
        raise ValueError("iq too short: %d < %d" % (iq.size, need))
    if mode in (MODE_DB10_U8, MODE_DB5_U8_DCFIX):
        out = np.empty((n_frames, n), dtype=np.uint8)
    elif mode == MODE_COMPLEX:
        out = np.empty((n_frames, n), dtype=np.complex128)
    else:
        out = np.empty((n_frames, n), dtype=np.float64)
    rc = lib().orc_rows(_u8p(iq), n_frames, n, hop, int(bool(flip)), mode,
                        out.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise ValueError("orc_rows failed: %d" % rc)
    return out


def rows_windowed(iq, n_frames, n, window, hop=None, flip=True, mode=MODE_MAG):
    """rows() with a taper in the weight slot of the unpack loop: x[j] = (-1)^j window[j] u8[j] / 256."""
    hop = n if hop is None else hop
    iq = np.ascontiguousarray(iq, dtype=np.uint8).ravel()
    need = 2 * ((n_frames - 1) * hop + n) if n_frames else 0
    if iq.size < need:
        raise ValueError("iq too short: %d < %d" % (iq.size, need))
    w = np.ascontiguousarray(window, dtype=np.float64)
    if w.size != n:
        raise ValueError("window must have n weights")
    if mode in (MODE_DB10_U8, MODE_DB5_U8_DCFIX):
        out = np.empty((n_frames, n), dtype=np.uint8)
    elif mode == MODE_COMPLEX:
        out = np.empty((n_frames, n), dtype=np.complex128)
    else:
        out = np.empty((n_frames, n), dtype=np.float64)
    L = lib()
    L.orc_rows_windowed.restype = ctypes.c_int
    L.orc_rows_windowed.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rc = L.orc_rows_windowed(iq.ctypes.data, n_frames, n, hop, int(bool(flip)), mode, w.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise ValueError("orc_rows_windowed failed: %d" % rc)
    return out


def host_threads():
    """Threads worth starting on this host: the affinity mask capped by the cgroup CPU quota (the GPU boxes show 256 logical
    CPUs under a quota of 16), at most 32."""
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cpus = min(cpus, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(32, cpus))


def rows_mt(iq, n_frames, n, hop=None, flip=True, mode=MODE_MAG, window=None, threads=None):
    """rows() / rows_windowed() with the frames sharded over `threads` pthreads (default: host_threads()): the same rows
    bit for bit, fast enough to check every row of a full-size configuration."""
    hop = n if hop is None else hop
    iq = np.ascontiguousarray(iq, dtype=np.uint8).ravel()
    need = 2 * ((n_frames - 1) * hop + n) if n_frames else 0
    if iq.size < need:
        raise ValueError("iq too short: %d < %d" % (iq.size, need))
    w = None
    if window is not None:
        w = np.ascontiguousarray(window, dtype=np.float64)
        if w.size != n:
            raise ValueError("window must have n weights")
    if mode in (MODE_DB10_U8, MODE_DB5_U8_DCFIX):
        out = np.empty((n_frames, n), dtype=np.uint8)
    elif mode == MODE_COMPLEX:
        out = np.empty((n_frames, n), dtype=np.complex128)
    else:
        out = np.empty((n_frames, n), dtype=np.float64)
    L = lib()
    L.orc_rows_mt.restype = ctypes.c_int
    L.orc_rows_mt.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                              ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    rc = L.orc_rows_mt(iq.ctypes.data, n_frames, n, hop, int(bool(flip)), mode, None if w is None else w.ctypes.data,
                       int(threads or host_threads()), out.ctypes.data)
    if rc != 0:
        raise ValueError("orc_rows_mt failed: %d" % rc)
    return out


WINDOW_KINDS = {"rect": 0, "boxcar": 0, "hann": 1, "hamming": 2, "blackman": 3, "blackmanharris": 4, "flattop": 5}


def window(kind, n):
    """Periodic cosine-sum taper of n weights (float64); kind: a name of WINDOW_KINDS or its number."""
    k = WINDOW_KINDS[kind] if isinstance(kind, str) else int(kind)
    w = np.empty(n, dtype=np.float64)
    L = lib()
    L.orc_window_fill.restype = ctypes.c_int
    L.orc_window_fill.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    if L.orc_window_fill(k, n, w.ctypes.data) != 0:
        raise ValueError("orc_window_fill(%r, %d) failed" % (kind, n))
    return w


def freq_shift(iq, freq_offset, sample_rate, state=(1.0, 0.0)):
    """nrf_freq_shifter_process on interleaved IQ (u8 -> u8/256.0, or f64): returns (interleaved f64
    output, new (cosine, sine) state)."""
    iq = np.ascontiguousarray(iq).ravel()
    c, s = ctypes.c_double(state[0]), ctypes.c_double(state[1])
    out = np.empty(iq.size, np.float64)
    if iq.dtype == np.uint8:
        lib().orc_freq_shift(_u8p(iq), None, iq.size // 2, freq_offset, sample_rate, ctypes.byref(c),
                             ctypes.byref(s), _f64p(out))
    else:
        iq = iq.astype(np.float64)
        lib().orc_freq_shift(None, _f64p(iq), iq.size // 2, freq_offset, sample_rate, ctypes.byref(c),
                             ctypes.byref(s), _f64p(out))
    return out, (c.value, s.value)


def rows_shifted(iq, n_frames, n, cycles_per_sample, phase0_cycles=0.0, hop=None, flip=True, mode=MODE_MAG):
    """rows() of the frequency-shifted stream (nrf_freq_shifter -> nrf_fft F64 branch)."""
    hop = n if hop is None else hop
    iq = np.ascontiguousarray(iq, dtype=np.uint8).ravel()
    need = 2 * ((n_frames - 1) * hop + n) if n_frames else 0
    if iq.size < need:
        raise ValueError("iq too short: %d < %d" % (iq.size, need))
    if mode in (MODE_DB10_U8, MODE_DB5_U8_DCFIX):
        out = np.empty((n_frames, n), dtype=np.uint8)
    elif mode == MODE_COMPLEX:
        out = np.empty((n_frames, n), dtype=np.complex128)
    else:
        out = np.empty((n_frames, n), dtype=np.float64)
    rc = lib().orc_rows_shifted(_u8p(iq), n_frames, n, hop, int(bool(flip)), mode, cycles_per_sample,
                                phase0_cycles, out.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise ValueError("orc_rows_shifted failed: %d" % rc)
    return out


def _rows_out(mode, n_frames, n):
    if mode in (MODE_DB10_U8, MODE_DB5_U8_DCFIX):
        return np.empty((n_frames, n), dtype=np.uint8)
    if mode == MODE_COMPLEX:
        return np.empty((n_frames, n), dtype=np.complex128)
    return np.empty((n_frames, n), dtype=np.float64)


def rows_shifted_windowed(iq, n_frames, n, cycles_per_sample, window, phase0_cycles=0.0, hop=None, flip=True, mode=MODE_MAG):
    """rows_shifted() with a taper beside the (-1)^n: x[j] = (-1)^j window[j] (shifter output)[j]."""
    hop = n if hop is None else hop
    iq = np.ascontiguousarray(iq, dtype=np.uint8).ravel()
    need = 2 * ((n_frames - 1) * hop + n) if n_frames else 0
    if iq.size < need:
        raise ValueError("iq too short: %d < %d" % (iq.size, need))
    w = np.ascontiguousarray(window, dtype=np.float64)
    if w.size != n:
        raise ValueError("window must have n weights")
    out = _rows_out(mode, n_frames, n)
    L = lib()
    L.orc_rows_shifted_windowed.restype = ctypes.c_int
    L.orc_rows_shifted_windowed.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
    rc = L.orc_rows_shifted_windowed(iq.ctypes.data, n_frames, n, hop, int(bool(flip)), mode, cycles_per_sample,
                                     phase0_cycles, w.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise ValueError("orc_rows_shifted_windowed failed: %d" % rc)
    return out


def rows_f64(x, n_frames, n, hop=None, mode=MODE_MAG, window=None):
    """Whole rows of nrf_fft_process' F64 branch (interleaved f64 IQ), with an optional taper beside the (-1)^n."""
    hop = n if hop is None else hop
    x = np.ascontiguousarray(x, dtype=np.float64).ravel()
    need = 2 * ((n_frames - 1) * hop + n) if n_frames else 0
    if x.size < need:
        raise ValueError("iq too short: %d < %d" % (x.size, need))
    w = None if window is None else np.ascontiguousarray(window, dtype=np.float64)
    if w is not None and w.size != n:
        raise ValueError("window must have n weights")
    out = _rows_out(mode, n_frames, n)
    L = lib()
    L.orc_rows_f64.restype = ctypes.c_int
    L.orc_rows_f64.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t, ctypes.c_int,
                               ctypes.c_void_p, ctypes.c_void_p]
    rc = L.orc_rows_f64(x.ctypes.data, n_frames, n, hop, mode, None if w is None else w.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise ValueError("orc_rows_f64 failed: %d" % rc)
    return out


def time_mag_rows(iq, n_frames, n, hop=None, threads=1):
    """cpu_baseline leg: seconds for n_frames reference-shaped rows."""
    hop = n if hop is None else hop
    iq = np.ascontiguousarray(iq, dtype=np.uint8).ravel()
    if threads <= 1:
        sink = np.empty(n, dtype=np.float64)
        return lib().orc_time_mag_rows(_u8p(iq), n_frames, n, hop, _f64p(sink))
    chk = ctypes.c_double(0)
    return lib().orc_time_mag_rows_mt(_u8p(iq), n_frames, n, hop, threads, ctypes.byref(chk))


def frequency_axis(image_width, image_height, rows_, fft_size, sample_rate, frequency_step, frequency_start,
                   frequency_end, minor_tick_rate=100000, major_tick_rate=1000000, line_color=255):
    """Ruler of the stitched image, numpy restatement of /root/reference/c/fft-stitch.c:56-72,191-217
    without the glyphs: returns (footer image [image_height][image_width] with banner lines and ticks,
    [(x, "%.2f" label)] for the major ticks that get a label).  Column 0 / row 0 are never written
    (img_pixel_put's guard); writes past the last row (the reference's first bottom banner line) are
    dropped."""
    img = np.zeros((image_height, image_width), np.uint8)

    def hline(y, x1, x2):
        if 0 < y < image_height:
            img[y, max(x1, 1):min(x2, image_width)] = line_color

    def vline(x, y1, y2):
        if 0 < x < image_width:
            img[max(y1, 1):min(y2, image_height), x] = line_color

    px_per_hz = fft_size / float(frequency_step) / 2
    minor, major = px_per_hz * minor_tick_rate, px_per_hz * major_tick_rate
    banner_y, banner_bottom = rows_, image_height
    for _ in range(10):
        hline(banner_y, 0, image_width)
        hline(banner_bottom, 0, image_width)
        banner_y += 1
        banner_bottom -= 1
    banner_bottom += 1
    x = 0.0
    while x < image_width:
        vline(int(x), banner_y, banner_y + 50)
        vline(int(x), banner_bottom - 50, banner_bottom)
        x += minor
    labels = []
    freq = frequency_start - sample_rate // 2 + major_tick_rate // 2
    x = fft_size / float(sample_rate) * (major_tick_rate // 2)
    while x < image_width:
        vline(int(x), banner_y, banner_y + 100)
        vline(int(x), banner_bottom - 100, banner_bottom)
        if 0 <= freq < frequency_end + sample_rate // 2:
            labels.append((int(x), "%.2f" % (freq / 1e6)))
        freq += major_tick_rate
        x += major
    return img, labels


def broad_markers(source, header, footer, frequency_start, frequency_end, sample_rate=5000000, footer_bleed=35,
                  minor_tick_rate=1000000, minor_tick_height=30, major_tick_rate=50000000, major_tick_height=60,
                  font_size_px=64, line_color=255):
    """Header + footer of the broad sweep poster, numpy restatement of /root/reference/c/add-markers.c:136-230
    without the glyphs (the geometry taken from `source` instead of the reference's asserted 23693 x 7157):
    returns (image [header + H + footer][W] with the source in place, border lines and ticks drawn,
    [(x, y, "%.2f" label)]).  img_pixel_put never writes column 0 / row 0 (c/add-markers.c:32-36) and
    img_vline skips x > stride (:38-43)."""
    h, w = source.shape
    out_h = header + h + footer
    img = np.zeros((out_h, w), np.uint8)
    img[header:header + h] = np.maximum(img[header:header + h], source)

    def hline(y, x1, x2):
        if 0 < y < out_h:
            img[y, max(x1, 1):min(x2, w)] = line_color

    def vline(x, y1, y2):
        if 0 < x < w:
            img[max(y1, 1):min(y2, out_h), x] = line_color

    real_start = frequency_start - sample_rate // 2
    real_end = frequency_end + sample_rate // 2
    real_range = real_end - real_start
    header_bottom, footer_top, footer_bottom = header, header + h, out_h - 1
    for i in range(10):
        hline(header_bottom - i, 0, w)
        hline(footer_top + i, 0, w)
    footer_top += 10

    def to_x(freq):
        if freq < real_start:
            return -1
        v = (freq - real_start) / float(real_range) * w
        return int(np.floor(v + 0.5)) if v >= 0 else -1                # C round(): half away from zero

    for freq in range(0, real_end, minor_tick_rate):
        x = to_x(freq)
        if 0 < x < w:
            for dx in (-1, 0, 1):
                vline(x + dx, footer_top, footer_top + minor_tick_height)
                vline(x + dx, footer_bottom - minor_tick_height - footer_bleed, footer_bottom + 1)
    labels = []
    labels_y = h + header + (footer // 2 - font_size_px // 2 - footer_bleed // 2)
    for freq in range(0, real_end, major_tick_rate):
        x = to_x(freq)
        if 0 < x < w:
            for dx in (-2, -1, 0, 1, 2):
                vline(x + dx, footer_top, footer_top + major_tick_height)
                vline(x + dx, footer_bottom - major_tick_height - footer_bleed, footer_bottom + 1)
            if real_start < freq < real_end:
                labels.append((x, labels_y, "%.2f" % (freq / 1e6)))
    return img, labels
