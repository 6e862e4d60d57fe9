"""GPU tier (-m gpu): the taper window fused into the kernels' byte conversion (fsea_plan_set_window; BASELINE.json's
"fused unpack+window prologue", config 5's STFT) against the oracle's windowed rows -- the reference's unpack loop
(src/nrf.c:601-614) with w[n] beside its (-1)^n.  Tolerances: tests/parity.py."""
import ctypes
import os

import numpy as np
import pytest

from frequensea_amd import fsea
from oracle import oracle as O
from tests import parity
from tests.conftest import GOLDEN_KEYS, ROOT, kernel_stem, synth_iq
from tests.test_gpu_parity import SIZES, DeviceBuffer, units_policy  # noqa: F401

pytestmark = pytest.mark.gpu


def _window(name, n, seed=0):
    if name == "random":
        return np.random.default_rng(seed + n).uniform(-1.0, 2.0, n).astype(np.float32)
    if name == "ones":
        return np.ones(n, np.float32)
    if name == "kaiser":
        from scipy.signal import get_window
        return get_window(("kaiser", 8.6), n).astype(np.float32)
    return fsea.window(name, n)


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("wname", ["hann", "blackmanharris", "random"])
def test_windowed_mag_rows_all_sizes(n, wname):
    nf = 300 if n <= 1024 else 37
    iq = synth_iq(7 * n + len(wname), 2 * nf * n)
    w = _window(wname, n)
    plan = fsea.Plan(n)
    plan.set_window(w)
    assert plan.window_form == (2 if wname == "random" else 1)
    assert plan.kernel_name == kernel_stem(n, 0) + "_u8_mag_win"
    got = plan.exec_host(iq, nf)
    parity.check_mode_windowed(got, iq, n, nf, n, True, 0, w)
    assert np.array_equal(got[:, n // 2], got[:, n // 2 - 1])
    plan.close()


@pytest.mark.parametrize("n", [32, 64, 256, 1024, 2048, 4096, 8192, 16384])
@pytest.mark.parametrize("mode", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("wname", ["hann", "random"])
def test_windowed_other_modes_and_byte_conventions(n, mode, wname):
    nf = 33
    iq = synth_iq(2000 + n + mode, 2 * nf * n)
    w = _window(wname, n, seed=mode)
    plan = fsea.Plan(n, mode=mode)
    plan.set_window(w)
    # raw int8 input: the pixel modes have compile-time windowed kernels, the others the run-time-mode one; offset-binary
    # input (flip = False) of the same plan runs the run-time-mode kernel -- same pixels
    assert plan.kernel_name == kernel_stem(n, mode) + {1: "_u8_db10_win", 2: "_u8_db5_win"}.get(mode, "_u8_win")
    for flip in (True, False):
        got = plan.exec_host(iq, nf, flip=flip)
        parity.check_mode_windowed(got, iq, n, nf, n, flip, mode, w)
    plan.close()


@pytest.mark.parametrize("n", SIZES)
def test_a_window_of_ones_gives_the_unwindowed_kernels_bits(n):
    """w == 1 is the reference itself: the windowed kernels (multiply by 1.0, DC term restored from the table) must give
    bit for bit what the un-windowed kernels give, in every mode."""
    nf = 64 if n <= 1024 else 17
    iq = synth_iq(31 * n, 2 * nf * n)
    for mode in (0, 1, 2, 3, 4, 5):
        plan = fsea.Plan(n, mode=mode)
        base = plan.exec_host(iq, nf)
        name = plan.kernel_name
        plan.set_window(np.ones(n, np.float32))
        assert plan.window_form == 1 and plan.kernel_name.endswith("_win")
        got = plan.exec_host(iq, nf)
        assert np.array_equal(base.view(np.uint8), got.view(np.uint8)), (n, mode)
        plan.set_window(None)                      # and the window can be taken off again
        assert plan.window_form == 0 and plan.kernel_name == name
        assert np.array_equal(base.view(np.uint8), plan.exec_host(iq, nf).view(np.uint8))
        plan.close()


def test_window_fill_matches_scipy_and_the_oracle():
    from scipy.signal import get_window
    for name in ("hann", "hamming", "blackman", "blackmanharris", "flattop"):
        for n in (32, 1024, 16384):
            w = fsea.window(name, n)
            assert np.allclose(w, get_window(name, n), rtol=0, atol=1e-7), name
            assert np.array_equal(w, O.window(name, n).astype(np.float32)), name
    assert np.array_equal(fsea.window("rect", 64), np.ones(64, np.float32))


def test_window_forms():
    """Cosine-sum tapers take the centred form (DC term restored from its spectrum around bin n/2); a taper whose spectrum
    is not confined to that band takes the offset-binary form; both inside the tolerance."""
    n, nf = 4096, 24
    iq = synth_iq(99, 2 * nf * n)
    plan = fsea.Plan(n)
    for name, form in (("hann", 1), ("hamming", 1), ("blackman", 1), ("flattop", 1), ("kaiser", 2), ("random", 2)):
        w = _window(name, n)
        plan.set_window(w)
        assert plan.window_form == form, name
        parity.check_mode_windowed(plan.exec_host(iq, nf), iq, n, nf, n, True, 0, w)
    plan.close()


def test_centred_form_keeps_weak_signals_accurate():
    """What the centred form is for: with a signal far below the offset-binary DC term (sigma 1.5 LSB) the error away from
    the DC term's bins stays small against the signal -- a Hann taper (centred form) against the same taper with a 1e-4
    ripple on it, which takes the offset-binary form and carries the DC term's f32 rounding noise into every bin."""
    n, nf = 8192, 16
    iq = synth_iq(5, 2 * nf * n, sigma=1.5, amp=3.0)
    keep = np.ones(n, bool)
    keep[n // 2 - 2: n // 2 + 3] = False            # everything but the DC term's five bins
    errs = {}
    plan = fsea.Plan(n, mode=fsea.MODE_COMPLEX_F32)
    ripple = (1.0 + 1e-4 * np.random.default_rng(2).standard_normal(n)).astype(np.float32)
    for name, w, form in (("hann", fsea.window("hann", n), 1), ("rippled", fsea.window("hann", n) * ripple, 2)):
        plan.set_window(w)
        assert plan.window_form == form
        got = plan.exec_host(iq, nf).astype(np.complex128)
        want = O.rows_windowed(iq, nf, n, w.astype(np.float64), mode=O.MODE_COMPLEX)
        parity.check_float(got, want)                # the stated tolerance holds for both
        errs[name] = np.linalg.norm((got - want)[:, keep]) / np.linalg.norm(want[:, keep])
    plan.close()
    assert errs["hann"] <= 3e-6 and errs["hann"] <= 0.5 * errs["rippled"], errs


@pytest.mark.parametrize("n", [8192, 16384])
def test_windowed_half_overlap_kernel(n, monkeypatch):
    """hop == n/2 with a window: the half-overlap kernel (every sample loaded once) against the oracle, and bit for bit
    against the ordinary windowed kernel."""
    hop, nf = n // 2, 301
    iq = synth_iq(n + 1, 2 * ((nf - 1) * hop + n))
    w = fsea.window("hann", n)
    plan = fsea.Plan(n, hop=hop)
    plan.set_window(w)
    assert plan.kernel_name == "fsea_fft%d_u8_mag_half_win" % n
    got = plan.exec_host(iq, nf)
    rows = np.r_[0:12, nf - 12:nf]
    for f in rows:
        parity.check_mode_windowed(got[f:f + 1], iq[2 * f * hop: 2 * (f * hop + n)], n, 1, n, True, 0, w)
    plan.close()
    monkeypatch.setenv("FSEA_NO_HALF_OVERLAP", "1")
    plain = fsea.Plan(n, hop=hop)
    plain.set_window(w)
    assert plain.kernel_name == "fsea_fft%d_u8_mag_win" % n
    assert np.array_equal(plain.exec_host(iq, nf), got)
    plain.close()


def test_config5_stft_with_hann_at_full_size():
    """BASELINE.json config 5 with the taper north_star names: 16384-point frames at hop 8192 over a 2^26-sample stream
    (8191 frames, device-resident), Hann.  Every row against the windowed oracle (frames sharded over the host's cores);
    the tone where it belongs in every row."""
    n, hop, nf = 16384, 8192, 8191
    n_samples = (nf - 1) * hop + n
    rng = np.random.default_rng(55)
    block = synth_iq(55, 2 * (1 << 20))
    reps = (2 * n_samples + block.size - 1) // block.size
    iq = np.tile(block, reps)[:2 * n_samples].copy()
    iq[::4097] ^= rng.integers(0, 8, iq[::4097].size).astype(np.uint8)   # the tiled blocks are not identical
    w = fsea.window("hann", n)
    plan = fsea.Plan(n, hop=hop)
    plan.set_window(w)
    d_in, d_out = DeviceBuffer(iq.nbytes).upload(iq), DeviceBuffer(nf * n * 4)
    plan.exec_device(d_in.ptr, nf, d_out.ptr)
    plan.synchronize()
    got = d_out.download(np.float32, (nf, n))
    w64 = np.asarray(w, np.float32).astype(np.float64)            # the weights the kernel applies are the f32 values
    for f0 in range(0, nf, 1024):
        cnt = min(1024, nf - f0)
        want = O.rows_mt(iq[2 * f0 * hop: 2 * ((f0 + cnt - 1) * hop + n)], cnt, n, hop=hop, window=w64)
        for g0 in range(0, cnt, 256):
            parity.check_float(got[f0 + g0: f0 + min(cnt, g0 + 256)], want[g0: g0 + 256])
    peak = np.argmax(got, axis=1)
    tone = n // 2 + n // 8
    dc_bins = {n // 2 - 1, n // 2, n // 2 + 1}
    assert all(int(p) == tone or int(p) in dc_bins for p in peak)
    d_in.free()
    d_out.free()
    plan.close()


@pytest.mark.parametrize("mode", [fsea.MODE_MAG_F32, fsea.MODE_DB5_U8_DCFIX])
def test_windowed_headline_batch_every_row(mode):
    """BASELINE config 3's batch (8192 points x 4096 frames) with a Hann taper, f32 rows and DB5 pixels: every row against the
    windowed oracle (frames sharded over the host's cores)."""
    n, nf = 8192, 4096
    iq = synth_iq(3, 2 * nf * n)
    w = fsea.window("hann", n)
    plan = fsea.Plan(n, mode=mode)
    plan.set_window(w)
    d_in = DeviceBuffer(iq.nbytes).upload(iq)
    d_out = DeviceBuffer(nf * n * (4 if mode == fsea.MODE_MAG_F32 else 1))
    plan.exec_device(d_in.ptr, nf, d_out.ptr)
    plan.synchronize()
    got = d_out.download(np.float32 if mode == fsea.MODE_MAG_F32 else np.uint8, (nf, n))
    want = O.rows_mt(iq, nf, n, mode=parity.ORACLE_MODE[mode], window=np.asarray(w, np.float32).astype(np.float64))
    for f0 in range(0, nf, 256):
        (parity.check_float if mode == fsea.MODE_MAG_F32 else parity.check_u8)(got[f0:f0 + 256], want[f0:f0 + 256])
    d_in.free()
    d_out.free()
    plan.close()


@pytest.mark.parametrize("n", [1024, 4096, 8192])
def test_windowed_long_launches_both_unit_distributions(n, units_policy):  # noqa: F811
    nf = 4096 if n >= 4096 else 8192
    iq = synth_iq(n * 3, 2 * nf * n)
    w = _window("blackman", n)
    plan = fsea.Plan(n)
    plan.set_window(w)
    d_in, d_out = DeviceBuffer(iq.nbytes).upload(iq), DeviceBuffer(nf * n * 4)
    plan.exec_device(d_in.ptr, nf, d_out.ptr)
    plan.synchronize()
    got = d_out.download(np.float32, (nf, n))
    for f in np.r_[0:6, nf // 2:nf // 2 + 4, nf - 6:nf]:
        parity.check_mode_windowed(got[f:f + 1], iq[2 * f * n: 2 * (f + 1) * n], n, 1, n, True, 0, w)
    plan.exec_device(d_in.ptr, nf, d_out.ptr)           # identical launches give identical rows
    plan.synchronize()
    assert np.array_equal(got, d_out.download(np.float32, (nf, n)))
    d_in.free()
    d_out.free()
    plan.close()


def test_windowed_tiles_equal_windowed_rows():
    n, tile_rows, n_tiles = 256, 64, 5
    nf = tile_rows * n_tiles
    iq = synth_iq(12, 2 * nf * n)
    w = fsea.window("hann", n)
    plan = fsea.Plan(n, mode=fsea.MODE_DB5_U8_DCFIX)
    plan.set_window(w)
    rows = plan.exec_host(iq, nf)
    d_in = DeviceBuffer(iq.nbytes).upload(iq)
    width = n * n_tiles + 64
    image = np.zeros((tile_rows, width), np.uint8)
    d_img = DeviceBuffer(image.nbytes).upload(image)
    plan.exec_tiled_device(d_in.ptr, nf, d_img.ptr, tile_rows, width, 32, tile_rows, n)
    plan.synchronize()
    image = d_img.download(np.uint8, image.shape)
    for k in range(n_tiles):
        assert np.array_equal(image[:, 32 + k * n: 32 + (k + 1) * n], rows[k * tile_rows:(k + 1) * tile_rows])
    d_in.free()
    d_img.free()
    plan.close()


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("wname", ["hann", "random"])
def test_windowed_frequency_shifted_rows(n, wname):
    """The nrf_freq_shifter -> nrf_fft chain (lua/fft-shifted.lua:52-55; src/nrf.c:843-866, 607-612) on a plan with a taper:
    x[j] = (-1)^j w[j] ((u8 / 256) e^{i phi} + 0.5 (1 + i)) -- the `*_u8_rot_win` kernels against the oracle's windowed
    shifted rows, every mode the run-time-mode kernel serves, both byte conventions, both forms of the DC term (cosine sum:
    the shifter's + 0.5 (1 + i) restored from the window's spectrum; arbitrary weights: carried through the transform)."""
    nf = 40 if n <= 1024 else 9
    iq = synth_iq(300 + n, 2 * nf * n)
    w = _window(wname, n, seed=3)
    cps, phase0 = 10e3 / 5e6 * (1 + n % 7), 0.3125
    for mode in (0, 3, 2):
        plan = fsea.Plan(n, mode=mode)
        plan.set_window(w)
        for flip in (True, False):
            got = plan.exec_shifted_host(iq, nf, cps, phase0, flip=flip)
            want = O.rows_shifted_windowed(iq, nf, n, cps, np.asarray(w, np.float32).astype(np.float64), phase0, flip=flip,
                                           mode=parity.ORACLE_MODE[mode])
            (parity.check_u8 if mode == 2 else parity.check_float)(got, want)
        if mode == 0:
            # a shift of zero still is the shifter: its + 0.5 (1 + i) comes on top of the offset-binary bytes' own DC term
            # and, under a taper, reaches the neighbours of bin n/2 -- the oracle's shifted rows at zero shift, not the
            # un-shifted windowed transform
            want0 = O.rows_shifted_windowed(iq, nf, n, 0.0, np.asarray(w, np.float32).astype(np.float64), 0.0)
            parity.check_float(plan.exec_shifted_host(iq, nf, 0.0), want0)
        plan.close()


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("wname", ["hann", "random"])
def test_windowed_f64_input_branch(n, wname):
    """nrf_fft_process' F64 branch (src/nrf.c:607-612: x[ii] = (-1)^ii f64) with the taper beside the sign: the `*_f32_win`
    kernels against the oracle, magnitude and complex rows; unit weights give the un-windowed kernel's bits."""
    nf = 25 if n <= 1024 else 6
    x = np.random.default_rng(n).normal(0.3, 0.2, 2 * nf * n)
    w = _window(wname, n, seed=5)
    w64 = np.asarray(w, np.float32).astype(np.float64)
    for mode in (0, 3):
        plan = fsea.Plan(n, mode=mode)
        base = plan.exec_host_f64(x, nf)
        plan.set_window(w)
        got = plan.exec_host_f64(x, nf)
        parity.check_float(got, O.rows_f64(x.astype(np.float32).astype(np.float64), nf, n, mode=parity.ORACLE_MODE[mode], window=w64))
        plan.set_window(np.ones(n, np.float32))
        assert np.array_equal(plan.exec_host_f64(x, nf).view(np.uint8), base.view(np.uint8)), (n, mode)
        plan.close()


def test_a_window_of_ones_gives_the_unwindowed_shifted_kernels_bits():
    for n in (256, 1024, 8192):
        nf = 12
        iq = synth_iq(77 + n, 2 * nf * n)
        plan = fsea.Plan(n)
        base = plan.exec_shifted_host(iq, nf, 0.0123, 0.4)
        plan.set_window(np.ones(n, np.float32))
        got = plan.exec_shifted_host(iq, nf, 0.0123, 0.4)
        # the DC term comes back from the table's f32 entry instead of the analytic 0.5 N (1 + i): the patched bin aside,
        # everything else is the same arithmetic
        assert np.array_equal(base, got), n
        plan.close()


@pytest.mark.parametrize("history", ["host", "device"])
@pytest.mark.parametrize("taper", ["hann", "blackmanharris"])
def test_nrf_fft_with_a_taper_from_the_environment(golden, monkeypatch, history, taper):
    """NRF_FFT_WINDOW (host/nrf_fft.c): the nrf_* API keeps its signatures and the taper reaches "the fft_buffer handed to Lua"
    -- U8 device blocks, F64 buffers (the nrf_freq_shifter -> nrf_fft chain, lua/fft-shifted.lua:52-55; src/nrf.c:607-612)
    and the shifter block itself, both history modes, at the sizes the scenes use -- against the windowed oracle."""
    from frequensea_amd import nrf
    monkeypatch.setenv("NRF_FFT_HISTORY", history)
    monkeypatch.setenv("NRF_FFT_WINDOW", taper)
    L = nrf.nrf_lib()
    raw = golden["rf_202p500_2__flipped"]                       # device buffers are offset binary
    for n, h in ((1024, 1024), (128, 512)):
        w = O.window(taper, n).astype(np.float32).astype(np.float64)
        fft = L.nrf_fft_new(n, h)
        buf = L.nut_buffer_new_u8(raw.size // 2, 2, raw.ctypes.data)
        L.nrf_fft_process(fft, buf)
        want_u8 = O.rows_windowed(raw, 1, n, w, flip=False)[0]
        f64 = L.nut_buffer_convert(buf, nrf.NUT_BUFFER_F64)
        L.nrf_fft_process(fft, f64)
        x = np.ctypeslib.as_array(f64.contents.data.f64, shape=(raw.size,)).copy()
        want_f64 = O.rows_f64(x.astype(np.float32).astype(np.float64), 1, n, window=w)[0]
        shifter = L.nrf_freq_shifter_new(10000, 5000000)
        L.nrf_freq_shifter_process(shifter, buf)
        sb = L.nrf_freq_shifter_get_buffer(shifter)
        L.nrf_fft_process(fft, sb)
        want_shift = O.rows_shifted_windowed(raw, 1, n, 10000 / 5e6, w, flip=False)[0]
        out = L.nrf_fft_get_buffer(fft)
        hist = nrf.buffer_to_numpy(L, out).reshape(h, n)
        L.nut_buffer_free(out)
        parity.check_float(hist[2], want_u8)
        parity.check_float(hist[1], want_f64)
        parity.check_float(hist[0], want_shift)
        assert not hist[3:].any()
        for b in (sb, f64, buf):
            L.nut_buffer_free(b)
        L.nrf_freq_shifter_free(shifter)
        L.nrf_fft_free(fft)
    # unset (or "rect"): the reference's rectangular frames, i.e. the golden rows
    monkeypatch.setenv("NRF_FFT_WINDOW", "rect")
    fft = L.nrf_fft_new(1024, 4)
    buf = L.nut_buffer_new_u8(raw.size // 2, 2, raw.ctypes.data)
    L.nrf_fft_process(fft, buf)
    out = L.nrf_fft_get_buffer(fft)
    parity.check_float(nrf.buffer_to_numpy(L, out).reshape(4, 1024)[0], golden["rf_202p500_2__mag_1024"])
    L.nut_buffer_free(out)
    L.nut_buffer_free(buf)
    L.nrf_fft_free(fft)


@pytest.mark.parametrize("history", ["host", "device"])
def test_two_nrf_fft_objects_choose_their_own_taper(golden, monkeypatch, history):
    """nrf_fft_set_window / nrf_fft_set_window_weights (include/nrf.h: additions beside the reference's five prototypes,
    VERDICT r05 item 4): two nrf_fft objects of one process carry different tapers at the same time, a third stays
    rectangular (the golden rows), the taper changes between process calls -- rows already in the history keep theirs --
    and the caller's own weights arrive bit for bit; NRF_FFT_WINDOW is only what a new block starts from."""
    from frequensea_amd import nrf
    monkeypatch.setenv("NRF_FFT_HISTORY", history)
    monkeypatch.delenv("NRF_FFT_WINDOW", raising=False)
    L = nrf.nrf_lib()
    raw = golden["rf_202p500_2__flipped"]                       # device buffers are offset binary
    n, h = 1024, 8
    buf = L.nut_buffer_new_u8(raw.size // 2, 2, raw.ctypes.data)
    ffts = {name: L.nrf_fft_new(n, h) for name in ("hann", "blackmanharris", "rect")}
    L.nrf_fft_set_window(ffts["hann"], b"hann")
    L.nrf_fft_set_window(ffts["blackmanharris"], b"blackmanharris")

    def newest(fft):
        out = L.nrf_fft_get_buffer(fft)
        hist = nrf.buffer_to_numpy(L, out).reshape(h, n)
        L.nut_buffer_free(out)
        return hist

    def want(name):
        if name == "rect":
            return O.rows(raw, 1, n, flip=False)[0]
        return O.rows_windowed(raw, 1, n, O.window(name, n).astype(np.float32).astype(np.float64), flip=False)[0]

    for rnd in range(2):                                        # interleaved: nothing of one object leaks into another
        for name, fft in ffts.items():
            L.nrf_fft_process(fft, buf)
        for name, fft in ffts.items():
            parity.check_float(newest(fft)[0], want(name))
    parity.check_float(newest(ffts["rect"])[0], golden["rf_202p500_2__mag_1024"])
    assert np.linalg.norm(newest(ffts["hann"])[0] - newest(ffts["blackmanharris"])[0]) > 1.0
    # a change between process calls: the next row has the new taper, the rows below keep the one they were computed with
    fft = ffts["hann"]
    L.nrf_fft_set_window(fft, b"flattop")
    L.nrf_fft_process(fft, buf)
    L.nrf_fft_set_window(fft, None)                             # NULL / "" / "rect" / "none": the reference's frames
    L.nrf_fft_process(fft, buf)
    ramp = (0.25 + np.arange(n) / n).astype(np.float32)         # not a cosine sum: the offset-binary form of the kernel
    L.nrf_fft_set_window_weights(fft, ramp.ctypes.data)
    L.nrf_fft_process(fft, buf)
    f64 = L.nut_buffer_convert(buf, nrf.NUT_BUFFER_F64)         # the F64 branch (src/nrf.c:607-612) with the caller's weights
    L.nrf_fft_process(fft, f64)
    hist = newest(fft)
    x = np.ctypeslib.as_array(f64.contents.data.f64, shape=(raw.size,)).copy()
    parity.check_float(hist[0], O.rows_f64(x.astype(np.float32).astype(np.float64), 1, n, window=ramp.astype(np.float64))[0])
    parity.check_float(hist[1], O.rows_windowed(raw, 1, n, ramp.astype(np.float64), flip=False)[0])
    parity.check_float(hist[2], want("rect"))
    parity.check_float(hist[3], want("flattop"))
    parity.check_float(hist[4], want("hann"))
    parity.check_float(hist[5], want("hann"))
    assert not hist[6:].any()
    L.nrf_fft_set_window_weights(fft, None)
    L.nrf_fft_process(fft, buf)
    assert np.array_equal(newest(fft)[0], newest(ffts["rect"])[0])   # back to the un-windowed kernel's bits
    # the environment is the default of NEW blocks only; an explicit call overrides it
    monkeypatch.setenv("NRF_FFT_WINDOW", "hamming")
    fresh = L.nrf_fft_new(n, h)
    L.nrf_fft_process(fresh, buf)
    parity.check_float(newest(fresh)[0], want("hamming"))
    L.nrf_fft_set_window(fresh, b"rect")
    L.nrf_fft_process(fresh, buf)
    assert np.array_equal(newest(fresh)[0], newest(ffts["rect"])[0])
    L.nrf_fft_free(fresh)
    for b in (f64, buf):
        L.nut_buffer_free(b)
    for fft in ffts.values():
        L.nrf_fft_free(fft)


def test_nrf_fft_set_window_refuses_an_unknown_name_loudly():
    import subprocess
    import sys
    code = ("from frequensea_amd import nrf; L = nrf.nrf_lib(); f = L.nrf_fft_new(1024, 4); "
            "L.nrf_fft_set_window(f, b'hamster'); print('not reached')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert r.returncode == 1 and 'nrf_fft_set_window: "hamster"' in r.stderr and "not reached" not in r.stdout


def test_nrf_fft_refuses_an_unknown_taper_loudly():
    """The reference's error convention (src/nrf.c:54-78): print and exit(EXIT_FAILURE)."""
    import subprocess
    import sys
    code = "from frequensea_amd import nrf; L = nrf.nrf_lib(); L.nrf_fft_new(1024, 4); print('not reached')"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT,
                       env=dict(os.environ, NRF_FFT_WINDOW="hamster"), timeout=300)
    assert r.returncode == 1 and "NRF_FFT_WINDOW=hamster" in r.stderr and "not reached" not in r.stdout
    r = subprocess.run([sys.executable, "-c", code.replace("1024, 4", "1000, 4")], capture_output=True, text=True, cwd=ROOT,
                       env=dict(os.environ, NRF_FFT_WINDOW="hann"), timeout=300)
    assert r.returncode == 1 and "no kernel of its own" in r.stderr and "not reached" not in r.stdout


def test_window_errors():
    plan = fsea.Plan(1024)
    bad = np.ones(1024, np.float32)
    bad[17] = np.nan
    with pytest.raises(fsea.FseaError, match="not finite"):
        plan.set_window(bad)
    plan.close()
    odd = fsea.Plan(1000, hop=1000)
    with pytest.raises(fsea.FseaError, match="no kernel of its own"):
        odd.set_window(np.ones(1000, np.float32))
    odd.close()


@pytest.mark.parametrize("key", GOLDEN_KEYS)
@pytest.mark.parametrize("n", [1024, 8192, 16384])
def test_recorded_captures_with_hann_match_golden(golden, key, n):
    """Hann-windowed rows of the reference's recorded captures against the committed scipy rows (tests/golden/make_golden.py:
    scipy.signal.get_window("hann", n) rounded to f32, scipy.fft in f64)."""
    plan = fsea.Plan(n)
    plan.set_window("hann")
    got = plan.exec_host(golden[key + "__raw"], 1)[0]
    parity.check_float(got, golden["%s__hann_mag_%d" % (key, n)])
    plan.close()
