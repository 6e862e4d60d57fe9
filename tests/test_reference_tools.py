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


# ---- the reference's FFT scenes under its own vendored Lua interpreter (oracle/_ref/lua_trace) ----
LUA_TRACER = os.path.join(ROOT, "oracle", "_ref", "lua_trace")
REF_LUA_DIR = "/root/reference/lua"


def _committed_traces():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "lua_scene_traces.json")) as fp:
        return json.load(fp)


def _num(v):
    return float(v) if isinstance(v, str) else v


def test_committed_lua_traces_are_what_the_oracle_computes():
    """tests/golden/lua_scene_traces.json without a GPU and without the reference: the five scenes (and this repository's own
    windowed scene, apart from them) are there, every call is
    one the replay knows, nrf_fft_shift's d is the float the binding hands over (src/main.cpp:788), and the checksums the
    trace recorded for every fft_buffer are the ones the oracle gives for the same call sequence on the same blocks -- if the
    oracle changes, the traces have to be regenerated (tests/golden/make_lua_traces.py)."""
    traces = _committed_traces()
    golden = np.load(os.path.join(ROOT, "tests", "golden", "rfdata_golden.npz"))
    blocks = []
    for key in traces["replay_blocks"]:
        blk = np.zeros(262144, np.uint8)
        blk[: golden[key + "__raw"].size] = golden[key + "__raw"]
        blocks.append(O.flip_u8(blk))
    assert sorted(traces["scenes"]) == ["fft-sea-auto.lua", "fft-sea-sick.lua", "fft-sea.lua", "fft-shifted.lua", "fft.lua"]
    known = {"nrf_device_new", "nrf_device_set_frequency", "nrf_device_get_samples_buffer", "nrf_fft_new", "nrf_fft_process",
             "nrf_fft_shift", "nrf_fft_get_buffer", "nrf_freq_shifter_new", "nrf_freq_shifter_process",
             "nrf_freq_shifter_get_buffer", "ngl_texture_update"}
    # this repository's own scene (tests/golden/scenes/fft-windowed.lua): kept apart from the reference's five, and the only
    # place where the taper addition of include/nrf.h may appear
    assert sorted(traces["own_scenes"]) == ["fft-windowed.lua"]
    for scene, body in list(traces["scenes"].items()) + list(traces["own_scenes"].items()):
        own = scene in traces["own_scenes"]
        ffts, shifters, bufs, windows = {}, {}, {}, {}
        seen = set()
        for ev in body["events"]:
            if ev["ev"] != "call":
                continue
            fn = ev["fn"]
            assert fn in known or (own and fn == "nrf_fft_set_window"), (scene, fn)
            seen.add(fn)
            if fn == "nrf_fft_set_window":
                name = ev["name"]
                windows[ev["fft"]] = None if name in ("rect", "none", "") else O.window(name, ffts[ev["fft"]][0]).astype(np.float32).astype(np.float64)
                continue
            if fn == "nrf_device_get_samples_buffer":
                bufs[ev["ret"]["id"]] = ("u8", blocks[ev["block"]])
            elif fn == "nrf_fft_new":
                ffts[ev["ret"]] = [ev["fft_size"], ev["fft_history_size"], np.zeros((ev["fft_history_size"], ev["fft_size"]))]
            elif fn == "nrf_fft_process":
                n, h, hist = ffts[ev["fft"]]
                kind, data = bufs[ev["buffer"]]
                w = windows.get(ev["fft"])
                if w is not None:
                    row = (O.rows_windowed(data[: 2 * n], 1, n, w, flip=False) if kind == "u8" else O.rows_f64(data[: 2 * n], 1, n, window=w))[0]
                else:
                    row = (O.rows(data[: 2 * n], 1, n, flip=False) if kind == "u8" else O.rows_f64(data[: 2 * n], 1, n))[0]
                ffts[ev["fft"]][2] = np.vstack([row[None, :], hist[:-1]])
            elif fn == "nrf_fft_shift":
                assert _num(ev["d"]) == np.float32(_num(ev["d_lua"]))
                n, h, hist = ffts[ev["fft"]]
                O.fft_shift(hist, n, h, _num(ev["d"]))
            elif fn == "nrf_fft_get_buffer":
                n, h, hist = ffts[ev["fft"]]
                assert ev["ret"]["length"] == n * h and ev["ret"]["channels"] == 1 and ev["ret"]["type"] == 2
                assert abs(hist.sum() - _num(ev["ret"]["sum"])) <= 1e-9 * max(1.0, _num(ev["ret"]["abs_sum"])), scene
                bufs[ev["ret"]["id"]] = ("hist", hist.copy())
            elif fn == "nrf_freq_shifter_new":
                shifters[ev["ret"]] = [ev["freq_offset"], ev["sample_rate"], (1.0, 0.0), None]
            elif fn == "nrf_freq_shifter_process":
                sh = shifters[ev["shifter"]]
                sh[3], sh[2] = O.freq_shift(bufs[ev["buffer"]][1], sh[0], sh[1], sh[2])
            elif fn == "nrf_freq_shifter_get_buffer":
                sh = shifters[ev["shifter"]]
                # src/nrf.c:853-856: the shifter's buffer counts VALUES as its length (twice the block, second half zero)
                assert ev["ret"]["length"] == 262144 and ev["ret"]["channels"] == 2
                assert abs(sh[3].sum() - _num(ev["ret"]["sum"])) <= 1e-9 * _num(ev["ret"]["abs_sum"])
                bufs[ev["ret"]["id"]] = ("f64", sh[3])
            elif fn == "ngl_texture_update":
                kind, hist = bufs[ev["buffer"]]
                assert kind == "hist" and ev["width"] * ev["height"] <= hist.size
                got = hist.ravel()[: ev["width"] * ev["height"]].astype(np.float32).sum(dtype=np.float64)
                assert abs(got - _num(ev["f32_sum"])) <= 1e-9 * max(1.0, abs(got))
        assert {"nrf_device_new", "nrf_fft_new", "nrf_fft_process", "nrf_fft_get_buffer", "nrf_fft_shift",
                "ngl_texture_update"} <= seen, scene
        assert ("nrf_fft_set_window" in seen) == own, scene          # the reference's scenes never call the addition
    # what the scripts do that a reading of them had missed (rounds 2-4 replayed a hand-written table)
    sea = [e for e in traces["scenes"]["fft-sea.lua"]["events"] if e["ev"] == "call" and e["fn"] == "nrf_fft_shift"]
    assert _num(sea[0]["d"]) == float("inf")                          # set_freq(freq) from setup(): d = 5 / 0
    auto = [e for e in traces["scenes"]["fft-sea-auto.lua"]["events"] if e["ev"] == "call" and e["fn"] == "nrf_fft_shift"]
    assert len(auto) >= 8 and _num(auto[1]["d"]) == 500.0 and _num(auto[1]["d_lua"]) != 500.0   # every draw; float narrowing
    assert any(e["ev"] == "call" and e["fn"] == "nrf_fft_shift" for e in traces["scenes"]["fft-sea-sick.lua"]["events"])
    # SURVEY 8(c)'s known answer, recorded from the reference's nrf.c: rf-100.900-1, N = 1024 -> sum of the row 1567.312172
    first = next(e for e in traces["scenes"]["fft.lua"]["events"] if e["ev"] == "call" and e["fn"] == "nrf_fft_get_buffer")
    assert abs(_num(first["ret"]["sum"]) - 1567.312172) < 1e-6


@pytest.mark.skipif(not (os.path.exists(LUA_TRACER) and os.path.isdir(REF_LUA_DIR)),
                    reason="needs oracle/_ref/lua_trace and /root/reference/lua (the build container)")
def test_the_committed_lua_traces_are_what_the_scripts_do_today(tmp_path):
    """Regenerates the traces from the reference's scripts and compares them with the committed file."""
    import json
    import sys
    env = dict(os.environ, PYTHONPATH=ROOT)
    script = os.path.join(ROOT, "tests", "golden", "make_lua_traces.py")
    code = ("import importlib.util, os, sys; spec = importlib.util.spec_from_file_location('m', %r); m = importlib.util.module_from_spec(spec);"
            "spec.loader.exec_module(m); m.HERE_OUT = %r; m.main()" % (script, str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    with open(tmp_path / "lua_scene_traces.json") as fp:
        fresh = json.load(fp)
    assert fresh == _committed_traces()
