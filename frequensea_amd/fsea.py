"""ctypes binding of libfsea_hip.so (include/fsea.h).  No compute happens here."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

UNITS_AUTO, UNITS_STATIC, UNITS_TICKETS = 0, 1, 2
MODE_MAG_F32 = 0
MODE_DB10_U8 = 1
MODE_DB5_U8_DCFIX = 2
MODE_COMPLEX_F32 = 3
MODE_MAG_NODC_F32 = 4
MODE_DB_F32 = 5
WINDOW_RECT, WINDOW_HANN, WINDOW_HAMMING, WINDOW_BLACKMAN, WINDOW_BLACKMANHARRIS, WINDOW_FLATTOP = range(6)
WINDOW_KINDS = {"rect": 0, "boxcar": 0, "hann": 1, "hamming": 2, "blackman": 3, "blackmanharris": 4, "flattop": 5}

_MODE_DTYPE = {
    MODE_MAG_F32: np.float32, MODE_DB10_U8: np.uint8, MODE_DB5_U8_DCFIX: np.uint8,
    MODE_COMPLEX_F32: np.complex64, MODE_MAG_NODC_F32: np.float32, MODE_DB_F32: np.float32,
}

# Every symbol include/fsea.h declares; tests check the built library exports all of them.
EXPORTS = [
    "fsea_device_count", "fsea_plan_create", "fsea_plan_destroy", "fsea_plan_reset", "fsea_plan_release_stream",
    "fsea_plan_grid", "fsea_plan_row_bytes", "fsea_plan_fft_size", "fsea_exec_u8_device", "fsea_exec_u8_tiled_device", "fsea_plan_set_unit_distribution",
    "fsea_exec_u8_host", "fsea_exec_f64_host", "fsea_exec_u8_shifted_device", "fsea_exec_u8_shifted_host", "fsea_mean_magnitude_u8_device",
    "fsea_composite_max_device", "fsea_stitch_tiles_device", "fsea_device_alloc", "fsea_device_free", "fsea_copy_to_device",
    "fsea_copy_to_host", "fsea_stream_synchronize", "fsea_host_alloc", "fsea_host_free",
    "fsea_plan_kernel_name", "fsea_last_error_string",
    "fsea_history_create", "fsea_history_destroy", "fsea_history_push_u8_host", "fsea_history_push_f64_host",
    "fsea_history_shift", "fsea_history_get_f64",
    "fsea_plan_set_window", "fsea_plan_window_form", "fsea_window_fill",
    "fsea_stream_create", "fsea_stream_destroy", "fsea_copy_to_device_async", "fsea_copy_to_host_async",
]
# include/fsea_tune.h: only libfsea_hip_tune.so (scripts/tune.py and friends) has these
TUNE_EXPORTS = ["fsea_plan_create_variant", "fsea_time_exec_u8_device", "fsea_time_exec_u8_rotating",
                "fsea_plan_read_trace", "fsea_tune_stream_1to2"]


class FseaError(RuntimeError):
    pass


_USE_TUNE = False


def use_tune_library():
    """Measurement scripts call this first: load libfsea_hip_tune.so (a superset build with kernel
    variants, ablations, traces and the timing helper) instead of the product library."""
    global _USE_TUNE
    if _LIB is not None and not _USE_TUNE:
        raise FseaError("use_tune_library() must be called before the library is first used")
    _USE_TUNE = True


def tune_lib_path():
    return os.path.join(_HERE, "libfsea_hip_tune.so")


def lib_path():
    # FSEA_HIP_LIB: load an alternative build of the same library
    if os.environ.get("FSEA_HIP_LIB"):
        return os.environ["FSEA_HIP_LIB"]
    return tune_lib_path() if _USE_TUNE else os.path.join(_HERE, "libfsea_hip.so")


def build(jobs=8):
    """Compile libfsea_hip.so and libfsea_nrf.so in-tree (hipcc cross-compiles gfx950)."""
    subprocess.check_call(["make", "-s", "-j%d" % jobs, "-C", os.path.join(_HERE, "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "host")])


_LIB = None


def hip_lib():
    """Load libfsea_hip.so; raises loudly if it was not built (no fallback)."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise FseaError("libfsea_hip.so is missing: run frequensea_amd.build() / __graft_entry__.build()")
        L = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        L.fsea_last_error_string.restype = ctypes.c_char_p
        L.fsea_device_count.argtypes = [ctypes.POINTER(ci)]
        L.fsea_plan_create.argtypes = [ctypes.POINTER(vp), ci, ci, ci, ci]
        L.fsea_plan_reset.argtypes = [vp]
        L.fsea_plan_release_stream.argtypes = [vp, vp]
        L.fsea_history_create.argtypes = [vp, ci, ctypes.POINTER(vp)]
        L.fsea_history_destroy.argtypes = [vp]
        L.fsea_history_push_u8_host.argtypes = [vp, vp, ci]
        L.fsea_history_push_f64_host.argtypes = [vp, vp]
        L.fsea_history_shift.argtypes = [vp, ci]
        L.fsea_history_get_f64.argtypes = [vp, vp]
        if hasattr(L, "fsea_plan_create_variant"):     # the tuning library
            L.fsea_plan_create_variant.argtypes = [ctypes.POINTER(vp), ci, ci, ci, ci, ctypes.c_char_p]
            L.fsea_time_exec_u8_device.argtypes = [vp, vp, sz, ci, vp, vp, ci, ctypes.POINTER(ctypes.c_float)]
            L.fsea_plan_read_trace.argtypes = [vp, vp, ctypes.c_uint]
            L.fsea_tune_stream_1to2.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp), ci, sz, ci, vp, ci,
                                                ctypes.POINTER(ctypes.c_float)]
            L.fsea_time_exec_u8_rotating.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ci, sz, ci, vp, ci,
                                                     ctypes.POINTER(ctypes.c_float)]
        L.fsea_plan_destroy.argtypes = [vp]
        L.fsea_plan_grid.argtypes = [vp, sz, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint),
                                     ctypes.POINTER(sz)]
        L.fsea_plan_row_bytes.argtypes = [vp]
        L.fsea_plan_row_bytes.restype = sz
        L.fsea_plan_fft_size.argtypes = [vp]
        L.fsea_plan_kernel_name.argtypes = [vp]
        L.fsea_plan_kernel_name.restype = ctypes.c_char_p
        L.fsea_exec_u8_device.argtypes = [vp, vp, sz, ci, vp, vp]
        L.fsea_exec_u8_tiled_device.argtypes = [vp, vp, sz, ci, vp, sz, sz, sz, sz, sz, vp]
        L.fsea_plan_set_unit_distribution.argtypes = [vp, ci]
        L.fsea_exec_u8_host.argtypes = [vp, vp, sz, ci, vp]
        L.fsea_exec_f64_host.argtypes = [vp, vp, sz, vp]
        L.fsea_exec_u8_shifted_device.argtypes = [vp, vp, sz, ci, ctypes.c_double, ctypes.c_double, vp, vp]
        L.fsea_exec_u8_shifted_host.argtypes = [vp, vp, sz, ci, ctypes.c_double, ctypes.c_double, vp]
        L.fsea_mean_magnitude_u8_device.argtypes = [vp, vp, sz, ci, ctypes.POINTER(ctypes.c_double), vp]
        L.fsea_composite_max_device.argtypes = [vp, vp] + [ctypes.c_uint32] * 7 + [ci, vp]
        L.fsea_stitch_tiles_device.argtypes = [vp, vp] + [ctypes.c_uint32] * 6 + [ci, vp]
        L.fsea_device_alloc.argtypes = [ci, sz, ctypes.POINTER(vp)]
        L.fsea_device_free.argtypes = [ci, vp]
        L.fsea_copy_to_device.argtypes = [ci, vp, vp, sz]
        L.fsea_copy_to_host.argtypes = [ci, vp, vp, sz]
        L.fsea_stream_synchronize.argtypes = [vp, vp]
        L.fsea_host_alloc.argtypes = [sz, ctypes.POINTER(vp)]
        L.fsea_host_free.argtypes = [vp]
        L.fsea_plan_set_window.argtypes = [vp, vp]
        L.fsea_plan_window_form.argtypes = [vp]
        L.fsea_window_fill.argtypes = [ci, ci, vp]
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise FseaError("fsea error %d: %s" % (rc, hip_lib().fsea_last_error_string().decode()))


def window(kind, n):
    """fsea_window_fill: the periodic cosine-sum taper `kind` (a name of WINDOW_KINDS or a number) as n float32 weights."""
    w = np.empty(n, dtype=np.float32)
    _check(hip_lib().fsea_window_fill(WINDOW_KINDS[kind] if isinstance(kind, str) else int(kind), n, w.ctypes.data))
    return w


def device_count():
    n = ctypes.c_int(0)
    hip_lib().fsea_device_count(ctypes.byref(n))
    return n.value


class Plan:
    """One (fft_size, hop, mode) plan on one device; thin wrapper over fsea_plan_*."""

    def __init__(self, fft_size, hop=None, mode=MODE_MAG_F32, device=0, variant=None):
        self._L = hip_lib()
        self._p = ctypes.c_void_p()
        self.fft_size = fft_size
        self.hop = fft_size if hop is None else hop
        self.mode = mode
        self.device = device
        # variant None: the product plan (its mode's configuration of the size); "" (tuning library): the size's first
        # configuration whatever the mode; a name: that tuning / per-mode configuration
        if variant is None or (variant == "" and not hasattr(self._L, "fsea_plan_create_variant")):
            _check(self._L.fsea_plan_create(ctypes.byref(self._p), fft_size, self.hop, mode, device))
        elif not hasattr(self._L, "fsea_plan_create_variant"):
            raise FseaError("kernel variants live in libfsea_hip_tune.so: call fsea.use_tune_library() first")
        else:
            _check(self._L.fsea_plan_create_variant(ctypes.byref(self._p), fft_size, self.hop, mode, device,
                                                    variant.encode()))

    def close(self):
        if self._p:
            self._L.fsea_plan_destroy(self._p)
            self._p = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def row_bytes(self):
        return self._L.fsea_plan_row_bytes(self._p)

    @property
    def out_dtype(self):
        return _MODE_DTYPE[self.mode]

    @property
    def kernel_name(self):
        return self._L.fsea_plan_kernel_name(self._p).decode()

    def grid(self, n_frames):
        g, b, l = ctypes.c_uint(0), ctypes.c_uint(0), ctypes.c_size_t(0)
        _check(self._L.fsea_plan_grid(self._p, n_frames, ctypes.byref(g), ctypes.byref(b), ctypes.byref(l)))
        return g.value, b.value, l.value

    def in_bytes(self, n_frames):
        return 2 * ((n_frames - 1) * self.hop + self.fft_size) if n_frames else 0

    def exec_device(self, d_iq_ptr, n_frames, d_out_ptr, flip=True, stream=0):
        """Device pointers (ints), asynchronous on `stream` (hipStream_t as int, 0 = null stream)."""
        _check(self._L.fsea_exec_u8_device(self._p, d_iq_ptr, n_frames, int(bool(flip)), d_out_ptr,
                                           stream or None))

    def exec_tiled_device(self, d_iq_ptr, n_frames, d_image_ptr, image_rows, image_stride, first_x, tile_rows, tile_step,
                          flip=True, stream=0):
        """Rows written straight into a stitched image (fsea_exec_u8_tiled_device): frame f = row f % tile_rows of
        tile f // tile_rows, tile k at columns first_x + k * tile_step."""
        _check(self._L.fsea_exec_u8_tiled_device(self._p, d_iq_ptr, n_frames, int(bool(flip)), d_image_ptr, image_rows,
                                                 image_stride, first_x, tile_rows, tile_step, stream or None))

    def set_window(self, w):
        """fsea_plan_set_window: fft_size float32 weights (a name of WINDOW_KINDS is filled in first), None removes it."""
        if w is None:
            _check(self._L.fsea_plan_set_window(self._p, None))
            return
        if isinstance(w, str):
            w = window(w, self.fft_size)
        wf = np.ascontiguousarray(w, dtype=np.float32).ravel()
        if wf.size != self.fft_size:
            raise ValueError("a window has fft_size weights")
        _check(self._L.fsea_plan_set_window(self._p, wf.ctypes.data))

    @property
    def window_form(self):
        """0 = no window, 1 = centred form, 2 = offset-binary form (include/fsea.h)."""
        return self._L.fsea_plan_window_form(self._p)

    def set_unit_distribution(self, policy):
        """UNITS_AUTO (default), UNITS_STATIC or UNITS_TICKETS: fsea_plan_set_unit_distribution."""
        _check(self._L.fsea_plan_set_unit_distribution(self._p, policy))

    def reset(self):
        _check(self._L.fsea_plan_reset(self._p))

    def release_stream(self, stream):
        """fsea_plan_release_stream: give back the counter slot `stream` holds (after its captured graphs are destroyed)."""
        _check(self._L.fsea_plan_release_stream(self._p, stream))

    def time_device(self, d_iq_ptr, n_frames, d_out_ptr, reps, flip=True, stream=0):
        """Tuning library only (fsea_time_exec_u8_device)."""
        if not hasattr(self._L, "fsea_time_exec_u8_device"):
            raise FseaError("the timing helper lives in libfsea_hip_tune.so: call fsea.use_tune_library() first")
        ms = ctypes.c_float(0)
        _check(self._L.fsea_time_exec_u8_device(self._p, d_iq_ptr, n_frames, int(bool(flip)), d_out_ptr,
                                                stream or None, reps, ctypes.byref(ms)))
        return ms.value

    def time_rotating(self, d_iq_ptrs, n_frames, d_out_ptrs, reps, flip=True, stream=0):
        """Tuning library only (fsea_time_exec_u8_rotating): launch i uses buffer set i % len(sets)."""
        n = len(d_iq_ptrs)
        ins = (ctypes.c_void_p * n)(*[p.value if isinstance(p, ctypes.c_void_p) else p for p in d_iq_ptrs])
        outs = (ctypes.c_void_p * n)(*[p.value if isinstance(p, ctypes.c_void_p) else p for p in d_out_ptrs])
        ms = ctypes.c_float(0)
        _check(self._L.fsea_time_exec_u8_rotating(self._p, ins, outs, n, n_frames, int(bool(flip)), stream or None, reps,
                                                  ctypes.byref(ms)))
        return ms.value

    def synchronize(self, stream=0):
        _check(self._L.fsea_stream_synchronize(self._p, stream or None))

    def exec_host(self, iq_u8, n_frames=None, flip=True):
        iq = np.ascontiguousarray(iq_u8, dtype=np.uint8).ravel()
        if n_frames is None:
            n_frames = 0 if iq.size < 2 * self.fft_size else (iq.size // 2 - self.fft_size) // self.hop + 1
        if iq.size < self.in_bytes(n_frames):
            raise ValueError("iq too short for %d frames" % n_frames)
        out = np.empty((n_frames, self.fft_size), dtype=self.out_dtype)
        _check(self._L.fsea_exec_u8_host(self._p, iq.ctypes.data, n_frames, int(bool(flip)), out.ctypes.data))
        return out

    def exec_host_into(self, iq_u8, n_frames, out, flip=True):
        """fsea_exec_u8_host into a caller-owned array (any host memory: pageable numpy, or pinned_array())."""
        if iq_u8.nbytes < self.in_bytes(n_frames) or out.nbytes < n_frames * self.row_bytes:
            raise ValueError("buffers too short for %d frames" % n_frames)
        _check(self._L.fsea_exec_u8_host(self._p, iq_u8.ctypes.data, n_frames, int(bool(flip)), out.ctypes.data))
        return out

    def exec_shifted_device(self, d_iq_ptr, n_frames, d_out_ptr, cycles_per_sample, phase0_cycles=0.0, flip=True,
                            stream=0):
        """fsea_exec_u8_shifted_device: the frequency shifter fused into the FFT's load."""
        _check(self._L.fsea_exec_u8_shifted_device(self._p, d_iq_ptr, n_frames, int(bool(flip)), cycles_per_sample,
                                                   phase0_cycles, d_out_ptr, stream or None))

    def exec_shifted_host(self, iq_u8, n_frames, cycles_per_sample, phase0_cycles=0.0, flip=True):
        iq = np.ascontiguousarray(iq_u8, dtype=np.uint8).ravel()
        if iq.size < self.in_bytes(n_frames):
            raise ValueError("iq too short for %d frames" % n_frames)
        out = np.empty((n_frames, self.fft_size), dtype=self.out_dtype)
        _check(self._L.fsea_exec_u8_shifted_host(self._p, iq.ctypes.data, n_frames, int(bool(flip)),
                                                 cycles_per_sample, phase0_cycles, out.ctypes.data))
        return out

    def exec_host_f64(self, iq_f64, n_frames):
        iq = np.ascontiguousarray(iq_f64, dtype=np.float64).ravel()
        if iq.size < self.in_bytes(n_frames):
            raise ValueError("iq too short for %d frames" % n_frames)
        out = np.empty((n_frames, self.fft_size), dtype=self.out_dtype)
        _check(self._L.fsea_exec_f64_host(self._p, iq.ctypes.data, n_frames, out.ctypes.data))
        return out

    def mean_magnitude_device(self, d_iq_ptr, n_frames, flip=True, stream=0):
        m = ctypes.c_double(0)
        _check(self._L.fsea_mean_magnitude_u8_device(self._p, d_iq_ptr, n_frames, int(bool(flip)),
                                                     ctypes.byref(m), stream or None))
        return m.value


class PinnedArray:
    """A numpy view of fsea_host_alloc'ed (page-locked) memory; .array is valid until close()."""

    def __init__(self, shape, dtype):
        self._ptr = ctypes.c_void_p()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        _check(hip_lib().fsea_host_alloc(n, ctypes.byref(self._ptr)))
        buf = (ctypes.c_char * n).from_address(self._ptr.value)
        self.array = np.frombuffer(buf, dtype=dtype).reshape(shape)

    def close(self):
        if self._ptr:
            self.array = None
            hip_lib().fsea_host_free(self._ptr)
            self._ptr = ctypes.c_void_p()


def composite_max_device(d_dst, d_src, dst_x, dst_y, width, height, dst_stride, dst_height, src_stride, device=0,
                         stream=0):
    _check(hip_lib().fsea_composite_max_device(d_dst, d_src, dst_x, dst_y, width, height, dst_stride, dst_height,
                                               src_stride, device, stream or None))


def stitch_tiles_device(d_image, d_tiles, n_tiles, first_x, width_step, width, height, image_stride, device=0,
                        stream=0):
    _check(hip_lib().fsea_stitch_tiles_device(d_image, d_tiles, n_tiles, first_x, width_step, width, height,
                                              image_stride, device, stream or None))
