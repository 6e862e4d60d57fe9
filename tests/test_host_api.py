"""CPU tier: the C host layer (libfsea_nrf.so) and the C-ABI library load and export every
symbol the headers declare; nut_buffer semantics are checked against the reference's own
src/nut.c (oracle/_ref/libnut_ref.so) when that build exists; the file-replay device needs no
GPU.  No GPU compute is called here."""
import ctypes
import os
import re

import numpy as np
import pytest

import frequensea_amd
from frequensea_amd import fsea, nrf
from oracle import oracle as O
from tests.conftest import ROOT, synth_iq


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b((?:fsea|nut|nrf)_[a-z0-9_]+)\s*\(", text))


def test_fsea_exports_every_declared_symbol():
    L = ctypes.CDLL(fsea.lib_path())
    names = declared_functions("fsea.h")
    assert names == set(fsea.EXPORTS)
    for name in names:
        assert hasattr(L, name), name


def test_window_fill_and_window_tables_without_a_gpu():
    """fsea_window_fill is host arithmetic (callable without a device): the periodic cosine-sum tapers are scipy's and the
    oracle's; a plan-less fsea_plan_set_window fails with FSEA_EINVAL, it does not crash."""
    from scipy.signal import get_window
    from oracle import oracle as O
    L = fsea.hip_lib()
    for name in ("hann", "hamming", "blackman", "blackmanharris", "flattop"):
        for n in (32, 1000, 8192):
            w = fsea.window(name, n)
            assert w.dtype == np.float32 and np.max(np.abs(w - get_window(name, n))) <= 1e-7, name
            assert np.array_equal(w, O.window(name, n).astype(np.float32)), name
    assert np.array_equal(fsea.window("rect", 100), np.ones(100, np.float32))
    assert L.fsea_window_fill(99, 8, np.zeros(8, np.float32).ctypes.data) == -1          # unknown kind: FSEA_EINVAL
    assert L.fsea_plan_set_window(None, None) == -1 and L.fsea_plan_window_form(None) == 0


def test_tuning_library_is_a_superset_and_the_product_has_no_variants():
    """include/fsea_tune.h lives in libfsea_hip_tune.so only: the product library registers one kernel
    set per size and exports no variant / trace / timing entry point."""
    import subprocess
    tune = ctypes.CDLL(fsea.tune_lib_path())
    names = declared_functions("fsea_tune.h")
    assert names == set(fsea.TUNE_EXPORTS)
    for name in names | set(fsea.EXPORTS):
        assert hasattr(tune, name), name
    prod = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(os.path.dirname(fsea.tune_lib_path()),
                                                                                "libfsea_hip.so")]).decode()
    for forbidden in ("fsea_abl", "variant", "read_trace", "time_exec", "8192v2", "8192r1"):
        assert forbidden not in prod, forbidden
    tuned = subprocess.check_output(["nm", "-D", "--defined-only", fsea.tune_lib_path()]).decode()
    assert "fsea_abl8192_io_u8_mag" in tuned and "fsea_fft8192v2_u8_mag" in tuned


def test_comm_library_exports_every_declared_symbol():
    """include/fsea_comm.h = libfsea_rccl.so (the multi-GPU gather: RCCL grouped send/recv)."""
    path = os.path.join(os.path.dirname(fsea.lib_path()), "libfsea_rccl.so")
    L = ctypes.CDLL(path)
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "fsea_comm.h")).read(), flags=re.S)
    names = set(re.findall(r"\b(fsea_comm_[a-z0-9_]+)\s*\(", text))
    assert names == {"fsea_comm_create", "fsea_comm_destroy", "fsea_comm_size", "fsea_comm_backend",
                     "fsea_comm_stream_create", "fsea_comm_stream_destroy", "fsea_comm_gather", "fsea_comm_barrier",
                     "fsea_comm_selftest_rccl", "fsea_comm_last_error"}
    for name in names:
        assert hasattr(L, name), name
    import subprocess
    deps = subprocess.check_output(["ldd", path]).decode()
    assert "librccl" in deps
    for lib in ("libfsea_hip.so", "libfsea_nrf.so"):               # the product libraries do not pull RCCL in
        assert "rccl" not in subprocess.check_output(["ldd", os.path.join(os.path.dirname(path), lib)]).decode()


def test_nrf_exports_every_declared_symbol():
    L = ctypes.CDLL(nrf.lib_path())
    for header, listed in (("nut.h", nrf.NUT_EXPORTS), ("nrf.h", nrf.NRF_EXPORTS + nrf.NRF_ADDITIONS)):
        names = {n for n in declared_functions(header) if not n.endswith("_fn")}
        assert names == set(listed), names ^ set(listed)
        for name in names:
            assert hasattr(L, name), name


def test_easypng_exports():
    L = ctypes.CDLL(nrf.lib_path())
    text = open(os.path.join(ROOT, "include", "easypng.h")).read()
    names = set(re.findall(r"\b((?:write|read)_gray_png\w*)\s*\(", re.sub(r"/\*.*?\*/", "", text, flags=re.S)))
    assert names == {"write_gray_png", "write_gray_png_chunked", "read_gray_png"}
    for name in names:
        assert hasattr(L, name)


def test_no_gpu_fails_loudly_without_fallback():
    if fsea.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(fsea.FseaError, match="no CPU fallback"):
        fsea.Plan(1024)


def test_plan_argument_validation_messages():
    # argument checks run before any device work, so they are testable without a GPU
    L = fsea.hip_lib()
    p = ctypes.c_void_p()
    for bad in (1, 0, -4, 1 << 21, 600000):                   # beyond four-step (2^20) and Bluestein (2n - 1 <= 2^20)
        assert L.fsea_plan_create(ctypes.byref(p), bad, 8 * max(1, abs(bad)), 0, 0) == -1
        assert b"unsupported fft_size" in L.fsea_last_error_string()
    for ok in (1000, 9000, 32768, 1 << 20, 524288, 17):      # Bluestein / four-step sizes: accepted, then no device here
        assert L.fsea_plan_create(ctypes.byref(p), ok, ok, 0, 0) == -2, ok
    assert L.fsea_plan_create(ctypes.byref(p), 1000, 0, 0, 0) == -1
    assert L.fsea_plan_create(ctypes.byref(p), 1024, 12, 0, 0) == -1
    assert b"multiple of 8" in L.fsea_last_error_string()
    assert L.fsea_plan_create(ctypes.byref(p), 1024, 1024, 9, 0) == -1
    assert L.fsea_plan_create(None, 1024, 1024, 0, 0) == -1
    assert L.fsea_plan_destroy(None) == 0


def _libs():
    ours = nrf.nrf_lib()
    path = os.path.join(ROOT, "oracle", "_ref", "libnut_ref.so")
    ref = nrf.bind_nut(ctypes.CDLL(path)) if os.path.exists(path) else None
    return ours, ref


def _drive_nut(L):
    """A fixed script of nut_buffer calls; returns everything observable."""
    seen = []
    u = np.arange(24, dtype=np.uint8) * 9
    f = np.linspace(-0.4, 0.95, 12)
    bu = L.nut_buffer_new_u8(12, 2, u.ctypes.data)
    bf = L.nut_buffer_new_f64(6, 2, f.ctypes.data)
    bz = L.nut_buffer_new_u8(5, 1, None)
    for b in (bu, bf, bz):
        c = b.contents
        seen.append((c.type, c.length, c.channels, c.size_bytes, nrf.buffer_to_numpy(L, b).tolist()))
    seen.append([L.nut_buffer_get_f64(bu, i) for i in range(24)])
    seen.append([L.nut_buffer_get_u8(bf, i) for i in range(12)])
    L.nut_buffer_set_f64(bu, 3, 0.503)
    L.nut_buffer_set_u8(bf, 2, 200)
    cp = L.nut_buffer_copy(bu)
    rd = L.nut_buffer_reduce(bu, 0.55)
    rd2 = L.nut_buffer_reduce(bf, 7.0)
    cl = L.nut_buffer_clip(bu, 4, 3)
    cl2 = L.nut_buffer_clip(bf, 2, -1)
    cv = L.nut_buffer_convert(bu, nrf.NUT_BUFFER_F64)
    cv2 = L.nut_buffer_convert(bf, nrf.NUT_BUFFER_U8)
    L.nut_buffer_append(cp, rd)
    five = L.nut_buffer_clip(bu, 7, 5)            # 5 elements x 2 channels, from element 7
    bz2 = L.nut_buffer_new_u8(5, 2, None)
    L.nut_buffer_set_data(bz2, five)
    seen.append(nrf.buffer_to_numpy(L, bz2).tolist())
    L.nut_buffer_free(five)
    L.nut_buffer_free(bz2)
    for b in (cp, rd, rd2, cl, cl2, cv, cv2, bz):
        c = b.contents
        seen.append((c.type, c.length, c.channels, c.size_bytes, nrf.buffer_to_numpy(L, b).tolist()))
    for b in (bu, bf, bz, cp, rd, rd2, cl, cl2, cv, cv2):
        L.nut_buffer_free(b)
    return seen


def test_nut_buffer_matches_reference_build():
    ours, ref = _libs()
    mine = _drive_nut(ours)
    # self-evident conventions (src/nut.c:121-151)
    assert mine[3][:3] == [0.0, 9 / 256.0, 18 / 256.0]
    if ref is None:
        pytest.skip("oracle/_ref/libnut_ref.so not built (reference absent)")
    assert mine == _drive_nut(ref)


def test_nut_buffer_save_roundtrip(tmp_path):
    ours, _ = _libs()
    data = np.arange(64, dtype=np.float64)
    b = ours.nut_buffer_new_f64(32, 2, data.ctypes.data)
    path = tmp_path / "buf.raw"
    ours.nut_buffer_save(b, str(path).encode())
    ours.nut_buffer_free(b)
    assert np.array_equal(np.fromfile(path, dtype=np.float64), data)


def test_replay_device_flips_and_steps(tmp_path):
    """nrf_device_new on a 3-block capture: samples = raw ^ 0x80 (src/nrf.c:100-109), paused
    device steps block by block and wraps (src/nrf.c:153-160, 341-350)."""
    L = nrf.nrf_lib()
    raw = synth_iq(11, 3 * nrf.NRF_BUFFER_SIZE_BYTES)
    path = tmp_path / "cap.raw"
    raw.tofile(path)
    dev = L.nrf_device_new(100.9, str(path).encode())
    try:
        L.nrf_device_set_paused(dev, 1)
        import time

        def current():
            time.sleep(0.06)          # > 2 replay periods of 1/60 s
            b = L.nrf_device_get_samples_buffer(dev)
            c = b.contents
            assert (c.type, c.length, c.channels, c.size_bytes) == (1, 131072, 2, 262144)
            arr = nrf.buffer_to_numpy(L, b)
            L.nut_buffer_free(b)
            return arr

        blocks = raw.reshape(3, -1) ^ np.uint8(0x80)
        first = current()
        idx = [i for i in range(3) if np.array_equal(first, blocks[i])]
        assert len(idx) == 1          # paused on exactly one block
        for step in range(1, 5):
            L.nrf_device_step(dev)
            assert np.array_equal(current(), blocks[(idx[0] + step) % 3])
        assert L.nrf_device_set_frequency(dev, 433.0) == 433.0
    finally:
        L.nrf_device_free(dev)


def test_replay_device_missing_file_is_zero_block():
    L = nrf.nrf_lib()
    dev = L.nrf_device_new(97.0, b"/nonexistent/capture.raw")
    try:
        import time
        time.sleep(0.05)
        b = L.nrf_device_get_samples_buffer(dev)
        arr = nrf.buffer_to_numpy(L, b)
        L.nut_buffer_free(b)
        assert np.all(arr == 0x80)    # int8 0 -> offset-binary 128
    finally:
        L.nrf_device_free(dev)


def test_png_writer_deflates_large_images_in_slabs(tmp_path):
    """Images of a MiB and more are deflated as up to sixteen slabs of rows on as many threads, concatenated into one zlib
    stream (host/easypng.c): ragged slab heights, compressible and incompressible content; decoded by PIL, by this
    library's reader and by zlib itself (one stream, correct Adler-32)."""
    import struct
    import zlib
    from PIL import Image
    L = ctypes.CDLL(nrf.lib_path())
    L.write_gray_png.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.read_gray_png.restype = ctypes.POINTER(ctypes.c_uint8)
    L.read_gray_png.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    rng = np.random.default_rng(5)
    for shape, kind in (((4099, 1024), "noise"), ((1031, 2048), "smooth"), ((8192, 256), "noise"), ((3, 700000), "smooth")):
        if kind == "noise":
            img = rng.integers(0, 256, shape, dtype=np.uint8)
        else:
            img = ((np.arange(shape[0])[:, None] // 7 + np.arange(shape[1])[None, :] // 5) % 251).astype(np.uint8)
        path = str(tmp_path / "big.png")
        assert L.write_gray_png(path.encode(), shape[1], shape[0], np.ascontiguousarray(img).ctypes.data) == 0
        with Image.open(path) as im:
            assert im.mode == "L" and np.array_equal(np.array(im), img)
        w, h = ctypes.c_int(), ctypes.c_int()
        p = L.read_gray_png(path.encode(), ctypes.byref(w), ctypes.byref(h))
        assert (h.value, w.value) == shape and np.array_equal(np.ctypeslib.as_array(p, shape=shape), img)
        data = open(path, "rb").read()
        pos, idat = 8, b""
        while pos < len(data):
            ln, typ = struct.unpack(">I4s", data[pos:pos + 8])
            if typ == b"IDAT":
                idat += data[pos + 8: pos + 8 + ln]
            pos += 12 + ln
        raw = zlib.decompress(idat)                      # checks the combined Adler-32 too
        assert len(raw) == shape[0] * (shape[1] + 1)


def test_png_writer_splits_the_compressed_data_into_idat_chunks(tmp_path):
    """A PNG chunk length is 31 bits, and the reference's own stitched image (154112 x 11811, c/fft-stitch.c:16-27) is
    1.8 GB of scanlines: the writer cuts the zlib stream into IDAT chunks (2^30 bytes by default, any size through
    write_gray_png_chunked), as libpng does for the reference.  Tiny chunk limits here, cutting inside the zlib header, the
    slabs and the Adler-32; decoded by PIL, by zlib chunk by chunk, and by this library's reader."""
    import struct
    import zlib
    from PIL import Image
    L = ctypes.CDLL(nrf.lib_path())
    L.write_gray_png_chunked.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
    L.read_gray_png.restype = ctypes.POINTER(ctypes.c_uint8)
    L.read_gray_png.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    rng = np.random.default_rng(6)
    for shape, limit in (((37, 256), 1), ((37, 256), 7), ((2100, 1024), 8192), ((2100, 1024), 1000003), ((5, 9), 1 << 40)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        path = str(tmp_path / "chunks.png")
        assert L.write_gray_png_chunked(path.encode(), shape[1], shape[0], img.ctypes.data, limit) == 0
        data = open(path, "rb").read()
        pos, lens, idat = 8, [], b""
        while pos < len(data):
            ln, typ = struct.unpack(">I4s", data[pos:pos + 8])
            assert zlib.crc32(data[pos + 4: pos + 8 + ln]) == struct.unpack(">I", data[pos + 8 + ln: pos + 12 + ln])[0]
            if typ == b"IDAT":
                lens.append(ln)
                idat += data[pos + 8: pos + 8 + ln]
            pos += 12 + ln
        assert pos == len(data) and max(lens) <= limit and all(x == min(limit, 1 << 30) for x in lens[:-1]) and lens[-1] > 0
        if limit < len(idat):
            assert len(lens) == -(-len(idat) // limit) > 1
        else:
            assert len(lens) == 1
        assert len(zlib.decompress(idat)) == shape[0] * (shape[1] + 1)
        with Image.open(path) as im:
            assert im.mode == "L" and np.array_equal(np.array(im), img)
        w, h = ctypes.c_int(), ctypes.c_int()
        p = L.read_gray_png(path.encode(), ctypes.byref(w), ctypes.byref(h))
        assert (h.value, w.value) == shape and np.array_equal(np.ctypeslib.as_array(p, shape=shape), img)


def test_png_writer_and_reader_against_pil(tmp_path):
    """include/easypng.h: same on-disk format as the reference's libpng writer (8-bit gray,
    non-interlaced); checked both ways against an independent codec (PIL)."""
    from PIL import Image
    L = ctypes.CDLL(nrf.lib_path())
    L.write_gray_png.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.read_gray_png.restype = ctypes.POINTER(ctypes.c_uint8)
    L.read_gray_png.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    for shape in ((1, 1), (37, 256), (300, 1024)):
        img = np.random.default_rng(shape[0]).integers(0, 256, shape, dtype=np.uint8)
        ours = str(tmp_path / "ours.png").encode()
        assert L.write_gray_png(ours, shape[1], shape[0], img.ctypes.data) == 0
        with Image.open(ours.decode()) as im:
            assert im.mode == "L" and np.array_equal(np.array(im), img)
        theirs = str(tmp_path / "pil.png")
        Image.fromarray(img).save(theirs)              # adaptive scanline filters
        w, h = ctypes.c_int(), ctypes.c_int()
        p = L.read_gray_png(theirs.encode(), ctypes.byref(w), ctypes.byref(h))
        assert (h.value, w.value) == shape
        assert np.array_equal(np.ctypeslib.as_array(p, shape=shape), img)
    rgb = np.random.default_rng(9).integers(0, 256, (8, 8, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(str(tmp_path / "rgb.png"))
    w, h = ctypes.c_int(), ctypes.c_int()
    p = L.read_gray_png(str(tmp_path / "rgb.png").encode(), ctypes.byref(w), ctypes.byref(h))
    want = ((rgb[..., 0].astype(int) * 77 + rgb[..., 1].astype(int) * 150 + rgb[..., 2].astype(int) * 29) >> 8)
    assert np.array_equal(np.ctypeslib.as_array(p, shape=(8, 8)), want.astype(np.uint8))
    assert not L.read_gray_png(b"/nonexistent.png", ctypes.byref(w), ctypes.byref(h))
    assert L.write_gray_png(b"/nonexistent-dir/x.png", 4, 4, img.ctypes.data) == -1


# ---------------------------------------------------------------------------------------------
# frequency ruler of the stitched image (include/imgaxis.h; c/fft-stitch.c:56-72,191-217)
# ---------------------------------------------------------------------------------------------
class _AxisCfg(ctypes.Structure):
    _fields_ = [("fft_size", ctypes.c_uint32), ("rows", ctypes.c_uint32), ("sample_rate", ctypes.c_uint32),
                ("frequency_step", ctypes.c_uint32), ("frequency_start", ctypes.c_uint64),
                ("frequency_end", ctypes.c_uint64), ("minor_tick_rate", ctypes.c_uint32),
                ("major_tick_rate", ctypes.c_uint32), ("font_size_px", ctypes.c_uint32),
                ("line_color", ctypes.c_uint8), ("font", ctypes.c_void_p)]


def _draw_axis(width, height, rows_, start, end, font_px, fft_size=1024, step=2000000):
    L = nrf.nrf_lib()
    L.img_draw_frequency_axis.restype = ctypes.c_int
    L.img_draw_frequency_axis.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(_AxisCfg)]
    cfg = _AxisCfg(fft_size, rows_, 5000000, step, start, end, 100000, 1000000, font_px, 255)
    img = np.zeros((height, width), np.uint8)
    n = L.img_draw_frequency_axis(img.ctypes.data, width, height, ctypes.byref(cfg))
    return img, n


def test_imgaxis_exports():
    import re
    text = open(os.path.join(ROOT, "include", "imgaxis.h")).read()
    names = re.findall(r"^(?:int|void)\s+(\w+)\(", text, flags=re.M)
    assert sorted(names) == ["img_draw_broad_markers", "img_draw_frequency_axis", "img_draw_text", "img_hline",
                             "img_pixel_put", "img_vline"]
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ntt_font.h")).read(), flags=re.S)
    font_names = set(re.findall(r"\b(ntt_font_[a-z_]+)\s*\(", text))
    assert {"ntt_font_load", "ntt_font_free", "ntt_font_measure", "ntt_font_draw", "ntt_font_glyph_bitmap"} <= font_names
    for name in font_names:
        assert hasattr(nrf.nrf_lib(), name), name
    for name in names:
        assert hasattr(nrf.nrf_lib(), name), name


@pytest.mark.parametrize("width,height,rows_,start,end", [
    (1024 + 4 * 512, 700, 100, 1802000000, 1810000000),       # the reference's 600-row footer
    (1024 + 299 * 512, 640, 40, 1802000000, 2400000000),      # the reference's full 154112-px sweep width
    (1024, 300, 64, 100000000, 100000000),
])
def test_axis_lines_and_ticks_match_the_restatement(width, height, rows_, start, end):
    got, n = _draw_axis(width, height, rows_, start, end, 0)
    want, labels = O.frequency_axis(width, height, rows_, 1024, 5000000, 2000000, start, end)
    assert np.array_equal(got, want)
    assert n == len(labels) and n > 0
    assert not got[:rows_].any()                       # the spectrogram rows are untouched
    assert not got[:, 0].any()                         # column 0 is never written (reference guard)
    # minor ticks every 25.6 px (truncated), 50 px long from the inner edge of both banners
    assert got[rows_ + 10, 25] == 255 and got[rows_ + 10, 51] == 255 and got[rows_ + 10, 26] == 0
    assert got[rows_ + 59, 51] == 255 and got[rows_ + 60, 51] == 0
    # first major tick at 1024 / 5e6 * 0.5e6 = 102.4 -> column 102, 100 px long
    assert got[rows_ + 109, 102] == 255 and got[rows_ + 110, 102] == 0


def test_axis_labels_sit_at_the_major_ticks():
    width, height, rows_ = 1024 + 4 * 512, 700, 100
    plain, _ = _draw_axis(width, height, rows_, 1802000000, 1810000000, 0)
    text, n = _draw_axis(width, height, rows_, 1802000000, 1810000000, 48)
    _, labels = O.frequency_axis(width, height, rows_, 1024, 5000000, 2000000, 1802000000, 1810000000)
    assert n == len(labels) and labels[0] == (102, "1800.00")
    extra = (text != plain)
    assert extra.any() and np.all(text[extra] == 255)
    ys, xs = np.nonzero(extra)
    markers_y = rows_ + (600 // 2 - 48 // 2)           # c/fft-stitch.c:36
    assert ys.min() >= markers_y and ys.max() < markers_y + 7 * (48 // 7)
    assert xs.min() >= labels[0][0]
    # "1800.00" starts with the glyph of '1': its 5x7 cell (scaled by 6) has the stem in column 2
    cell = 48 // 7
    assert extra[markers_y + 3 * cell, 102 + 2 * cell]


class _MarkersCfg(ctypes.Structure):
    _fields_ = [("source_height", ctypes.c_uint32), ("header_height", ctypes.c_uint32), ("footer_height", ctypes.c_uint32),
                ("footer_bleed", ctypes.c_uint32), ("sample_rate", ctypes.c_uint32), ("frequency_start", ctypes.c_uint64),
                ("frequency_end", ctypes.c_uint64), ("minor_tick_rate", ctypes.c_uint32),
                ("minor_tick_height", ctypes.c_uint32), ("major_tick_rate", ctypes.c_uint32),
                ("major_tick_height", ctypes.c_uint32), ("font_size_px", ctypes.c_uint32), ("line_color", ctypes.c_uint8),
                ("font", ctypes.c_void_p)]


@pytest.mark.parametrize("width,rows_,start,end,major", [
    (256 * 21, 200, 600000000, 700000000, 50000000),       # 21 tiles of 5 MHz: two 50 MHz labels inside
    (23693, 64, 50000000, 6000000000, 50000000),           # the reference poster's width and frequency range
    (256, 40, 100000000, 100000000, 1000000),              # a single tile
])
def test_broad_markers_match_the_restatement(width, rows_, start, end, major):
    """include/imgaxis.h img_draw_broad_markers vs the numpy restatement of c/add-markers.c:136-230."""
    L = nrf.nrf_lib()
    L.img_draw_broad_markers.restype = ctypes.c_int
    L.img_draw_broad_markers.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(_MarkersCfg)]
    src = np.random.default_rng(width).integers(0, 200, (rows_, width), dtype=np.uint8)
    want, labels = O.broad_markers(src, 300, 300, start, end, major_tick_rate=major)
    got = np.zeros_like(want)
    got[300:300 + rows_] = src
    cfg = _MarkersCfg(rows_, 300, 300, 35, 5000000, start, end, 1000000, 30, major, 60, 0, 255)
    n = L.img_draw_broad_markers(got.ctypes.data, width, ctypes.byref(cfg))
    assert np.array_equal(got, want)
    assert n == len(labels)
    assert np.array_equal(got[301:300 + rows_], src[1:])              # the image rows below the border line are untouched
    assert got[291:301, 1:].min() == 255 and got[300 + rows_: 310 + rows_, 1:].min() == 255     # ten border lines each
    # with labels: only pixels inside the label band change, all to the line colour
    cfg.font_size_px = 64
    text = got.copy()
    L.img_draw_broad_markers(text.ctypes.data, width, ctypes.byref(cfg))
    extra = text != got
    if labels:
        ys, xs = np.nonzero(extra)
        assert extra.any() and ys.min() >= labels[0][1] and ys.max() < labels[0][1] + 64 and xs.min() >= labels[0][0]
    else:
        assert not extra.any()


def test_add_markers_tool_end_to_end(tmp_path):
    """fsea-add-markers (c/add-markers.c main): broad-stitched-S-E.png -> broad-stitched-S-E-markers.png."""
    import subprocess
    L = nrf.nrf_lib()
    L.write_gray_png.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.read_gray_png.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    src = np.random.default_rng(3).integers(0, 255, (77, 256 * 11), dtype=np.uint8)
    assert L.write_gray_png(str(tmp_path / "broad-stitched-630-680.png").encode(), src.shape[1], src.shape[0],
                            src.ctypes.data) == 0
    tool = os.path.join(ROOT, "frequensea_amd", "bin", "fsea-add-markers")
    out = subprocess.run([tool, "--start", "630", "--end", "680", "--dir", str(tmp_path)], capture_output=True, text=True,
                         check=True).stdout
    assert "Adding markers..." in out
    w, h = ctypes.c_int(0), ctypes.c_int(0)
    L.read_gray_png.restype = ctypes.POINTER(ctypes.c_uint8)
    p = L.read_gray_png(str(tmp_path / "broad-stitched-630-680-markers.png").encode(), ctypes.byref(w), ctypes.byref(h))
    got = np.ctypeslib.as_array(p, shape=(h.value, w.value)).copy()
    want, labels = O.broad_markers(src, 300, 300, 630000000, 680000000)
    assert got.shape == want.shape == (677, 2816)
    band = np.zeros(got.shape, bool)
    for x, y, _ in labels:
        band[y:y + 64, x:] = True
    assert labels == [(1152, 300 + 77 + 101, "650.00")]               # (650 - 627.5) / 55 MHz * 2816 px
    assert np.array_equal(got[~band], want[~band]) and (got[band] != want[band]).any()
    assert subprocess.run([tool, "--start", "1", "--end", "2", "--dir", str(tmp_path)], capture_output=True).returncode != 0


def test_dot_matrix_text_advance_and_clipping():
    L = nrf.nrf_lib()
    L.img_draw_text.restype = ctypes.c_int
    L.img_draw_text.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int,
                                ctypes.c_int, ctypes.c_int, ctypes.c_uint8]
    img = np.zeros((20, 40), np.uint8)
    assert L.img_draw_text(img.ctypes.data, 40, 20, b"10.5", 2, 3, 7, 200) == 4 * 6
    assert img.max() == 200 and img[3:10, 2:7].any() and not img[:3].any()
    img2 = np.zeros((20, 40), np.uint8)
    assert L.img_draw_text(img2.ctypes.data, 40, 20, b"888888888", 30, 15, 14, 9) == 9 * 12   # runs off both edges
    assert img2[15:, 30:].any()
    assert L.img_draw_text(img2.ctypes.data, 40, 20, b"1", -1, 0, 7, 9) == 0                 # c/fft-stitch.c:130


# ---------------------------------------------------------------------------------------------
# nrf_freq_shifter (src/nrf.c:817-870), the block in front of nrf_fft in lua/fft-shifted.lua
# ---------------------------------------------------------------------------------------------
def test_freq_shifter_block_matches_the_restatement():
    L = nrf.nrf_lib()
    iq = synth_iq(31, 2 * 4096) ^ np.uint8(0x80)
    sh = L.nrf_freq_shifter_new(150000, 5000000)
    try:
        assert not L.nrf_freq_shifter_get_buffer(sh)              # nothing processed yet
        state = (1.0, 0.0)
        for block in range(3):                                     # the phase carries over blocks
            buf = L.nut_buffer_new_u8(4096, 2, iq.ctypes.data)
            L.nrf_freq_shifter_process(sh, buf)
            L.nut_buffer_free(buf)
            out = L.nrf_freq_shifter_get_buffer(sh)
            c = out.contents
            # the reference sizes its output by values, not samples (src/nrf.c:851): twice the room
            assert (c.type, c.length, c.channels, c.size_bytes) == (2, 8192, 2, 8192 * 2 * 8)
            got = nrf.buffer_to_numpy(L, out)
            L.nut_buffer_free(out)
            want, state = O.freq_shift(iq, 150000, 5000000, state)
            assert np.array_equal(got[:8192], want)                 # same recurrence, same doubles
            assert not got[8192:].any()
        # F64 input: values are used as they are
        f = (iq[:512].astype(np.float64) / 256.0)
        buf = L.nut_buffer_new_f64(256, 2, f.ctypes.data)
        L.nrf_freq_shifter_process(sh, buf)
        L.nut_buffer_free(buf)
        out = L.nrf_freq_shifter_get_buffer(sh)
        got = nrf.buffer_to_numpy(L, out)[:512]
        L.nut_buffer_free(out)
        want, state = O.freq_shift(f, 150000, 5000000, state)
        assert np.array_equal(got, want)
    finally:
        L.nrf_freq_shifter_free(sh)


def test_freq_shifter_process_samples_rotates_in_place():
    L = nrf.nrf_lib()
    sh = L.nrf_freq_shifter_new(-250000, 2000000)
    try:
        rng = np.random.default_rng(5)
        i, q = rng.normal(size=1000), rng.normal(size=1000)
        i0, q0 = i.copy(), q.copy()
        L.nrf_freq_shifter_process_samples(sh, i.ctypes.data, q.ctypes.data, 600)
        L.nrf_freq_shifter_process_samples(sh, i[600:].ctypes.data, q[600:].ctypes.data, 400)
        want = (i0 + 1j * q0) * np.exp(2j * np.pi * (-250000 / 2000000) * np.arange(1000))
        assert np.abs(i + 1j * q - want).max() < 1e-12           # no 0.5 offset on this entry point
    finally:
        L.nrf_freq_shifter_free(sh)
