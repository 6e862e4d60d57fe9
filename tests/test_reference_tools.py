"""The reference's own stitch tool, compiled as is from /root/reference/c/fft-stitch-broad.c into
oracle/_ref/ (oracle/Makefile), pins three things that otherwise rest on restatements only:
the max-composite a14 (oracle.composite_max), the tile/stitched-image file format (include/easypng.h
writer and reader against the reference's libpng writer and stb_image reader), and the file naming.
Skipped when the binary is absent (no reference tree / no png.h when the oracle was built)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from frequensea_amd import nrf
from oracle import oracle as O
from tests.conftest import ROOT

REF_STITCH_BROAD = os.path.join(ROOT, "oracle", "_ref", "fft-stitch-broad")


def _png_io():
    L = ctypes.CDLL(nrf.lib_path())
    L.write_gray_png.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.read_gray_png.restype = ctypes.POINTER(ctypes.c_uint8)
    L.read_gray_png.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    return L


def _read(L, path):
    w, h = ctypes.c_int(), ctypes.c_int()
    p = L.read_gray_png(str(path).encode(), ctypes.byref(w), ctypes.byref(h))
    assert p, path
    return np.ctypeslib.as_array(p, shape=(h.value, w.value)).copy()


def run_reference_stitch_broad(tmp_path, start, end):
    """Runs the reference binary in tmp_path (it reads broad-<MHz>.png from the cwd) and returns the
    path of the image it wrote."""
    try:
        out = subprocess.run([REF_STITCH_BROAD, str(start), str(end)], cwd=tmp_path, capture_output=True, text=True,
                             timeout=300)
    except OSError as e:                                       # e.g. built for another loader / libc
        pytest.skip("oracle/_ref/fft-stitch-broad cannot be executed here: %s" % e)
    if out.returncode in (126, 127) or "error while loading shared libraries" in out.stderr:
        pytest.skip("oracle/_ref/fft-stitch-broad cannot be executed here: %s" % out.stderr.strip())
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Image size: %d x 4096" % (256 + (end - start) // 5 * 256) in out.stdout
    return tmp_path / ("broad-stitched-%d-%d.png" % (start, end))


@pytest.mark.skipif(not os.path.exists(REF_STITCH_BROAD), reason="oracle/_ref/fft-stitch-broad not built")
def test_reference_stitch_broad_pins_composite_and_png_format(tmp_path):
    L = _png_io()
    rng = np.random.default_rng(77)
    freqs = [660, 665, 670]
    tiles = {}
    for k, f in enumerate(freqs):
        # taller than FFT_HISTORY_SIZE = 4096: the reference uses the first 4096 rows (c/fft-stitch-broad.c:74-82)
        t = rng.integers(0, 256, (4096 + 3 * k, 256), dtype=np.uint8)
        assert L.write_gray_png(str(tmp_path / ("broad-%d.png" % f)).encode(), 256, t.shape[0], t.ctypes.data) == 0
        tiles[f] = t
    ref_png = run_reference_stitch_broad(tmp_path, 660, 670)       # stb_image decoded our tiles
    got = _read(L, ref_png)                                        # our reader decodes libpng's file
    from PIL import Image
    with Image.open(ref_png) as im:
        assert im.mode == "L" and np.array_equal(np.array(im), got)
    want = np.zeros((4096, 3 * 256), np.uint8)
    for k, f in enumerate(freqs):
        O.composite_max(want, np.ascontiguousarray(tiles[f][:4096]), k * 256)   # WIDTH_STEP = 256 / (5e6 / 5e6)
    assert np.array_equal(got, want)


@pytest.mark.skipif(not os.path.exists(REF_STITCH_BROAD), reason="oracle/_ref/fft-stitch-broad not built")
def test_reference_stitch_broad_rejects_what_the_restatement_rejects(tmp_path):
    """Error paths the host tool mirrors (c/fft-stitch-broad.c:70-78): a missing tile, a tile of the
    wrong width."""
    L = _png_io()
    t = np.zeros((4096, 256), np.uint8)
    assert L.write_gray_png(str(tmp_path / "broad-100.png").encode(), 256, 4096, t.ctypes.data) == 0
    out = subprocess.run([REF_STITCH_BROAD, "100", "105"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 1 and "could not load broad-105.png" in out.stderr
    bad = np.zeros((4096, 128), np.uint8)
    assert L.write_gray_png(str(tmp_path / "broad-105.png").encode(), 128, 4096, bad.ctypes.data) == 0
    out = subprocess.run([REF_STITCH_BROAD, "100", "105"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 1 and "bad image size broad-105.png" in out.stderr
