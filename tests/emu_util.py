"""Builds and drives tests/emu/libfsea_emu.so: the real kernel source compiled for the CPU
(TEST-ONLY; see tests/emu/hip/hip_runtime.h)."""
import ctypes
import fcntl
import os
import subprocess

import numpy as np

from tests.conftest import ROOT

_EMU = None
OUT_DTYPE = {0: np.float32, 1: np.uint8, 2: np.uint8, 3: np.complex64, 4: np.float32, 5: np.float32}


def emu_lib():
    global _EMU
    if _EMU is None:
        emu = os.path.join(ROOT, "tests", "emu")
        srcs = [os.path.join(emu, f) for f in ("emu_main.cpp", "emu_variants_a.cpp", "emu_variants_b.cpp", "emu_variants_c.cpp")]
        out = os.path.join(emu, "libfsea_emu.so")
        csrc = os.path.join(ROOT, "frequensea_amd", "csrc")
        deps = srcs + [os.path.join(emu, "emu_common.h"), os.path.join(emu, "hip", "hip_runtime.h")] + [
            os.path.join(csrc, f) for f in ("fsea_fft_core.h", "fsea_fft_tune_members.h", "fsea_opt.h", "fsea_configs.h",
                                             "fsea_configs_tune.h", "fsea_tables.h")] + [
            os.path.join(emu, "fsea_pk_asm.h"), os.path.join(emu, "fsea_pk_asm_tune.h")]
        def stale():
            return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)

        if stale():
            # pytest-xdist workers may get here together: one builds, the others wait for the lock
            with open(out + ".lock", "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                if stale():
                    from concurrent.futures import ThreadPoolExecutor
                    flags = ["g++", "-std=c++20", "-O1", "-fPIC", "-pthread", "-Wno-unknown-pragmas", "-DFSEA_TUNE", "-I" + emu, "-I" + csrc]
                    objs = ["%s.%d.o" % (s_[:-4], os.getpid()) for s_ in srcs]

                    def compile_one(pair):
                        subprocess.check_call(flags + ["-c", pair[0], "-o", pair[1]])
                    with ThreadPoolExecutor(len(srcs)) as pool:       # the four translation units compile in parallel
                        list(pool.map(compile_one, zip(srcs, objs)))
                    tmp = "%s.%d.tmp" % (out, os.getpid())
                    subprocess.check_call(["g++", "-shared", "-pthread", "-o", tmp] + objs)
                    for o in objs:
                        os.remove(o)
                    os.replace(tmp, out)
        L = ctypes.CDLL(out)
        L.emu_fft.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                              ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_uint]
        L.emu_fft_variant.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_uint]
        _EMU = L
    return _EMU


def emu_rows(iq, n, n_frames, hop=None, flip=True, mode=0, grid=2, specialised=True, in_kind=0, variant="",
             shift=None, dynamic_units=True, run_len=None, window=None, window_mode=2, window_form=0):
    """shift = (cycles_per_sample, phase0_cycles) selects the frequency-shifted u8 kernel (in_kind 2);
    dynamic_units=False runs the multi-wave sizes with the static unit interleave (FftArgs::dynamic_units = 0);
    window = n float weights runs the windowed kernels (window_mode = FftKernel's WIN: 1 weights fetched per frame, 2
    register-resident; window_form 2 forces the offset-binary form)."""
    hop = n if hop is None else hop
    emu_lib().emu_set_window.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    if window is not None:
        wf = np.ascontiguousarray(window, dtype=np.float32)
        assert wf.size == n
        emu_lib().emu_set_window(wf.ctypes.data, n, int(window_mode), int(window_form))
    else:
        emu_lib().emu_set_window(None, 0, 2, 0)
    if run_len is not None:                      # the half-overlap MAG kernel (hop == n / 2), runs of run_len frames
        in_kind = 3
        emu_lib().emu_set_run_len.argtypes = [ctypes.c_uint32]
        emu_lib().emu_set_run_len(int(run_len))
    emu_lib().emu_set_dynamic_units.argtypes = [ctypes.c_uint32]
    emu_lib().emu_set_dynamic_units(1 if dynamic_units else 0)
    if shift is not None:
        in_kind = 2
        emu_lib().emu_set_shift.argtypes = [ctypes.c_double, ctypes.c_double]
        emu_lib().emu_set_shift(float(shift[0]), float(shift[1]))
    src = np.ascontiguousarray(iq)
    out = np.zeros((n_frames, n), dtype=OUT_DTYPE[mode])
    rc = emu_lib().emu_fft_variant(n, variant.encode(), in_kind, int(specialised), src.ctypes.data,
                                   out.ctypes.data, n_frames, hop, int(bool(flip)), mode, grid)
    assert rc == 0, "emu_fft_variant(%d, %r) = %d" % (n, variant, rc)
    return out


def emu_tiled(iq, n, n_frames, image_shape, first_x, tile_rows, tile_step, mode=0, grid=2, specialised=True, variant="",
              fill=0):
    """The kernel's tiled-output addressing (fsea_exec_u8_tiled_device): rows written into an image of `image_shape`
    pre-filled with `fill`."""
    L = emu_lib()
    L.emu_set_tiles.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_size_t]
    image = np.full(image_shape, fill, dtype=OUT_DTYPE[mode])
    L.emu_set_tiles(tile_rows, image_shape[1], tile_step, image.size - first_x)
    src = np.ascontiguousarray(iq)
    rc = L.emu_fft_variant(n, variant.encode(), 0, int(specialised), src.ctypes.data,
                           image.ctypes.data + first_x * image.itemsize, n_frames, n, 1, mode, grid)
    assert rc == 0
    return image
