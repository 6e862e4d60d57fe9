"""CPU: pin the C oracle against the golden vectors, analytic known answers and
the values SURVEY.md 8(c) recorded from the reference's own nrf.c."""
import ctypes
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import ALL_CAPTURE_KEYS, GOLDEN_KEYS, GOLDEN_SIZES, ROOT, synth_iq


def test_flip_identity_all_bytes():
    b = np.arange(256, dtype=np.uint8)
    assert np.array_equal(O.flip_u8(b), b ^ 0x80)          # (b+128)%256 == b^0x80
    assert np.array_equal(O.flip_u8(b), ((b.astype(int) + 128) % 256).astype(np.uint8))


@pytest.mark.parametrize("key", GOLDEN_KEYS)
def test_flip_golden(golden, key):
    assert np.array_equal(O.flip_u8(golden[key + "__raw"]), golden[key + "__flipped"])


def test_survey_known_answers(golden):
    """values reproduced from the reference's nrf.c."""
    raw = golden["rf_100p900_1__raw"]
    assert list(O.flip_u8(raw[:8])) == [113, 148, 103, 117, 119, 111, 140, 123]
    row = O.rows(raw, 1, 1024)[0]
    np.testing.assert_allclose(row[:4], [0.155516, 0.194782, 0.178461, 0.141473], atol=5e-7)
    assert row.argmax() == 513
    assert abs(row.max() - 37.935861) < 1e-6
    assert abs(row.sum() - 1567.312172) < 1e-6
    spec = O.rows(raw, 1, 1024, mode=O.MODE_COMPLEX)[0]
    assert abs(abs(spec[512]) - 728.425422) < 1e-6
    row = O.rows(raw, 1, 8192)[0]
    assert row.argmax() == 3358 and abs(row.max() - 162.437851) < 1e-6
    assert abs(row.sum() - 29851.896936) < 1e-5
    row = O.rows(golden["rf_202p500_2__raw"], 1, 1024)[0]
    assert row.argmax() == 666 and abs(row.max() - 255.674368) < 1e-6
    assert abs(row.sum() - 1669.595959) < 1e-6


@pytest.mark.parametrize("key", GOLDEN_KEYS)
@pytest.mark.parametrize("n", GOLDEN_SIZES)
def test_rows_match_golden(golden, key, n):
    raw = golden[key + "__raw"]
    want = golden["%s__mag_%d" % (key, n)]
    got = O.rows(raw, 1, n)[0]
    # two independent double FFTs: agreement to ~1e-12 relative to the row scale
    assert np.max(np.abs(got - want)) <= 1e-11 * max(1.0, want.max())
    assert got[n // 2] == got[n // 2 - 1]
    for mode, name in [(O.MODE_DB10_U8, "db10"), (O.MODE_DB5_U8_DCFIX, "db5")]:
        wantp = golden["%s__%s_%d" % (key, name, n)]
        gotp = O.rows(raw, 1, n, mode=mode)[0]
        diff = np.abs(gotp.astype(int) - wantp.astype(int))
        assert diff.max() <= 1 and np.count_nonzero(diff) <= max(1, n // 1000)


@pytest.mark.parametrize("key", ALL_CAPTURE_KEYS)
def test_rows_of_every_recorded_capture_match_golden(golden_all, key):
    """All 35 full-size captures of the reference's rfdata/ ("the same rfdata/*.raw inputs", BASELINE.json)."""
    raw = golden_all[key + "__raw"]
    for n in (1024, 8192):
        want = golden_all["%s__mag_%d" % (key, n)]
        got = O.rows(raw, 1, n)[0]
        assert np.max(np.abs(got - want)) <= 1e-11 * max(1.0, want.max())


def test_all_35_full_size_captures_are_in_the_fixture(golden_all):
    assert len(ALL_CAPTURE_KEYS) == 35 and "rf_433p000_short" not in ALL_CAPTURE_KEYS
    assert all(golden_all[k + "__raw"].size == 2 * 8192 for k in ALL_CAPTURE_KEYS)


def test_whole_block_of_one_capture_matches_golden(golden_all):
    """The 128 consecutive 1024-point frames of one 262144-byte block (one libhackrf transfer, c/fft-batch.c:54)."""
    raw = golden_all["block__raw"]
    assert raw.size == 262144
    got = O.rows(raw, 128, 1024)
    want = golden_all["block__mag_1024"]
    assert np.max(np.abs(got - want)) <= 1e-11 * want.max()


@pytest.mark.parametrize("key", GOLDEN_KEYS)
@pytest.mark.parametrize("n", [1024, 8192, 16384])
def test_hann_rows_match_golden(golden, key, n):
    """The windowed oracle (orc_rows_windowed, orc_window_fill) against scipy's window and scipy's FFT."""
    raw = golden[key + "__raw"]
    w = O.window("hann", n).astype(np.float32).astype(np.float64)
    got = O.rows_windowed(raw, 1, n, w)[0]
    want = golden["%s__hann_mag_%d" % (key, n)]
    assert np.max(np.abs(got - want)) <= 1e-11 * max(1.0, want.max())


def test_oracle_windows_are_scipys():
    from scipy.signal import get_window
    for name in ("hann", "hamming", "blackman", "blackmanharris", "flattop"):
        for n in (32, 1000, 1024, 16384):
            assert np.max(np.abs(O.window(name, n) - get_window(name, n))) <= 1e-14, name
    assert np.array_equal(O.window("rect", 64), np.ones(64))


def test_windowed_oracle_is_the_plain_one_for_unit_weights_and_linear_in_the_window():
    rng = np.random.default_rng(3)
    n, nf = 512, 3
    raw = rng.integers(0, 256, 2 * n * nf).astype(np.uint8)
    for mode in (O.MODE_MAG, O.MODE_DB10_U8, O.MODE_DB5_U8_DCFIX, O.MODE_COMPLEX):
        assert np.array_equal(O.rows_windowed(raw, nf, n, np.ones(n), mode=mode), O.rows(raw, nf, n, mode=mode))
    w1, w2 = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    a = O.rows_windowed(raw, nf, n, w1, mode=O.MODE_COMPLEX)
    b = O.rows_windowed(raw, nf, n, w2, mode=O.MODE_COMPLEX)
    c = O.rows_windowed(raw, nf, n, 2 * w1 - 3 * w2, mode=O.MODE_COMPLEX)
    assert np.max(np.abs(c - (2 * a - 3 * b))) <= 1e-9
    # a Hann taper is the 3-tap stencil 0.5 X[k] - 0.25 X[k-1] - 0.25 X[k+1] of the rectangular spectrum
    x = O.rows(raw, nf, n, mode=O.MODE_COMPLEX)
    h = O.rows_windowed(raw, nf, n, O.window("hann", n), mode=O.MODE_COMPLEX)
    assert np.max(np.abs(h - (0.5 * x - 0.25 * np.roll(x, 1, axis=1) - 0.25 * np.roll(x, -1, axis=1)))) <= 1e-9


@pytest.mark.parametrize("key", GOLDEN_KEYS[:2])
@pytest.mark.parametrize("n", [128, 256, 1024])
def test_spectrum_matches_golden_and_naive(golden, key, n):
    raw = golden[key + "__raw"]
    spec = O.rows(raw, 1, n, mode=O.MODE_COMPLEX)[0]
    want = golden["%s__spec_%d" % (key, n)]
    assert np.max(np.abs(spec - want)) <= 1e-11 * np.abs(want).max()
    x = O.unpack_center_u8(O.flip_u8(raw[: 2 * n]))
    naive = O.dft_naive(x)
    assert np.max(np.abs(spec - naive)) <= 1e-11 * np.abs(naive).max()


def test_unpack_center_exact():
    u = np.arange(256, dtype=np.uint8).repeat(2)          # I=Q=k for sample k
    x = O.unpack_center_u8(u)
    k = np.arange(256)
    sign = np.where(k % 2 == 0, 1.0, -1.0)
    assert np.array_equal(x.real, sign * k / 256.0) and np.array_equal(x.imag, sign * k / 256.0)
    f = np.linspace(-1, 1, 64)
    y = O.unpack_center_f64(f)
    assert np.array_equal(y.real, f[0::2] * np.where(np.arange(32) % 2 == 0, 1, -1))


@pytest.mark.parametrize("n", [2, 4, 8, 64, 512, 2048])
def test_fft_analytic(n):
    rng = np.random.default_rng(n)
    x = rng.normal(size=n) + 1j * rng.normal(size=n)
    X = O.fft_forward(x)
    assert np.allclose(X, np.fft.fft(x), rtol=0, atol=1e-11 * n)
    # impulse -> flat spectrum
    imp = np.zeros(n, complex); imp[0] = 0.75
    assert np.allclose(O.fft_forward(imp), 0.75)
    # Parseval
    assert abs(np.sum(np.abs(X) ** 2) - n * np.sum(np.abs(x) ** 2)) < 1e-9 * n * n
    # linearity
    y = rng.normal(size=n) + 1j * rng.normal(size=n)
    assert np.allclose(O.fft_forward(2 * x - 3j * y), 2 * X - 3j * O.fft_forward(y), atol=1e-10 * n)
    with pytest.raises(ValueError):
        O.fft_forward(np.zeros(3, complex))


def test_constant_input_dc_patch():
    n = 1024
    raw = np.zeros(2 * n, dtype=np.uint8)                  # int8 0 -> u8 128 after flip
    spec = O.rows(raw, 1, n, mode=O.MODE_COMPLEX)[0]
    assert abs(abs(spec[n // 2]) - 0.5 * n * np.sqrt(2)) < 1e-9
    others = np.delete(np.abs(spec), n // 2)
    assert others.max() < 1e-9
    row = O.rows(raw, 1, n)[0]
    assert row.max() < 1e-9                                # DC never reaches the output


def test_tone_lands_at_shifted_bin():
    n, k0 = 1024, 100
    t = np.arange(n)
    z = 40 * np.exp(2j * np.pi * k0 * t / n)
    raw = np.empty(2 * n, dtype=np.int8)
    raw[0::2] = np.rint(z.real); raw[1::2] = np.rint(z.imag)
    row = O.rows(raw.view(np.uint8), 1, n)[0]
    assert row.argmax() == (k0 + n // 2) % n


def test_history_scroll_and_shift(golden):
    hist = golden["shift__history"].copy()
    n, h = hist.shape[1], hist.shape[0]
    scrolled = hist.copy()
    O.history_scroll(scrolled, n, h)
    assert np.array_equal(scrolled[1:], hist[:-1]) and np.array_equal(scrolled[0], hist[0])
    for name in ["p8", "m8", "half", "p50", "m3", "big", "mhalf"]:
        d = float(golden["shift__" + name + "__d"][0])
        got = hist.copy()
        O.fft_shift(got, n, h, d)
        assert np.array_equal(got, golden["shift__" + name]), name


def test_db_rows_truncate_and_clamp():
    spec = np.array([0, 1, 10, 1e3, 1e13, 10 ** 1.26, 10 ** (25.5 / 20)], dtype=complex)
    px = O.db_u8_row(spec, 10.0, 0)
    # 0 -> 10*log10(1e-20)*10 = -2000 -> 0 ; 1 -> 0 ; 10 -> 200 ; 1e3 -> 600 -> 255
    assert list(px[:5]) == [0, 0, 200, 255, 255]
    assert px[5] == 252 and px[6] in (254, 255)
    px5 = O.db_u8_row(np.array([1, 10, 100, 1000], complex), 5.0, 1)
    assert list(px5) == [0, 100, 100, 255]               # pixel n/2 (=2) copies pixel 1


def test_mean_magnitude_gate():
    spec = np.full(100 * 256, 1.0 + 0j)
    assert abs(O.mean_magnitude(spec) - 1.0) < 1e-12     # < 1.1 -> "not interesting"
    assert O.mean_magnitude(spec * 2) > 1.1


def test_composite_max():
    dst = np.zeros((4, 12), np.uint8)
    a = np.arange(32, dtype=np.uint8).reshape(4, 8)
    O.composite_max(dst, a, 0)
    O.composite_max(dst, (31 - a).astype(np.uint8), 4)     # 50 % overlap
    want = np.zeros((4, 12), np.uint8)
    want[:, :8] = a
    want[:, 4:] = np.maximum(want[:, 4:], 31 - a)
    assert np.array_equal(dst, want)


def test_reference_nut_conventions():
    """oracle/_ref/libnut_ref.so is the reference's own src/nut.c (built by
    oracle/Makefile when /root/reference exists): u8<->f64 is /256.0, *256.0."""
    path = os.path.join(ROOT, "oracle", "_ref", "libnut_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (reference absent)")
    L = ctypes.CDLL(path)
    L.nut_buffer_new_u8.restype = ctypes.c_void_p
    L.nut_buffer_new_u8.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.nut_buffer_get_f64.restype = ctypes.c_double
    L.nut_buffer_get_f64.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.nut_buffer_free.argtypes = [ctypes.c_void_p]
    data = np.arange(256, dtype=np.uint8)
    buf = L.nut_buffer_new_u8(128, 2, data.ctypes.data)
    for k in (0, 1, 128, 255):
        assert L.nut_buffer_get_f64(buf, k) == k / 256.0
    L.nut_buffer_free(buf)


# ---------------------------------------------------------------------------------------------
# frequency shifter in front of the FFT (src/nrf.c:843-866; lua/fft-shifted.lua:52-55)
# ---------------------------------------------------------------------------------------------
def test_freq_shift_recurrence_against_closed_form():
    iq = synth_iq(71, 2 * 131072) ^ 0x80
    out, state = O.freq_shift(iq, 150000, 5000000)
    m = np.arange(131072)
    rot = np.exp(2j * np.pi * (150000 / 5000000) * m)
    v = (iq[0::2] / 256.0 + 1j * (iq[1::2] / 256.0)) * rot + (0.5 + 0.5j)
    assert np.abs(out[0::2] + 1j * out[1::2] - v).max() < 1e-10
    # the state carries the phase across blocks: two half blocks == one block
    a, st = O.freq_shift(iq[:131072], 150000, 5000000)
    b, st = O.freq_shift(iq[131072:], 150000, 5000000, st)
    assert np.abs(np.concatenate([a, b]) - out).max() < 1e-12
    assert abs(st[0] - state[0]) < 1e-12 and abs(st[1] - state[1]) < 1e-12
    # f64 input branch (nut_buffer_get_f64 of an F64 buffer is the value itself)
    c, _ = O.freq_shift(iq[:4096].astype(np.float64) / 256.0, 150000, 5000000)
    assert np.abs(c - out[:4096]).max() < 1e-15


@pytest.mark.parametrize("n", [256, 1024, 8192])
def test_rows_shifted_is_shifter_then_fft(n):
    nf, hop = 3, n
    raw = synth_iq(72 + n, 2 * nf * n)
    delta = 150000 / 5000000
    got = O.rows_shifted(raw, nf, n, delta, mode=O.MODE_COMPLEX)
    shifted, _ = O.freq_shift(O.flip_u8(raw), 150000, 5000000)       # the reference's own chain
    for f in range(nf):
        x = O.unpack_center_f64(shifted[2 * f * hop: 2 * (f * hop + n)])
        assert np.abs(O.fft_forward(x) - got[f]).max() < 1e-9 * n
    mag = O.rows_shifted(raw, nf, n, delta, mode=O.MODE_MAG)
    want = np.abs(got)
    want[:, n // 2] = want[:, n // 2 - 1]
    assert np.allclose(mag, want, rtol=0, atol=1e-12 * n)


def test_rows_shifted_by_whole_bins_rotates_the_spectrum():
    n, nf, k = 1024, 2, 37
    raw = synth_iq(73, 2 * nf * n)
    plain = O.rows(raw, nf, n, mode=O.MODE_COMPLEX)
    dc = np.zeros(n, complex)
    dc[n // 2] = 0.5 * n * (1 + 1j)
    # the unshifted stream already carries 0.5 (1+i) of offset-binary DC in bin n/2; the shifter moves
    # that with everything else and then adds a new 0.5 (1+i)
    got = O.rows_shifted(raw, nf, n, k / n, mode=O.MODE_COMPLEX)
    assert np.abs(got - (np.roll(plain, k, axis=1) + dc)).max() < 1e-8
    zero = O.rows_shifted(raw, nf, n, 0.0, mode=O.MODE_COMPLEX)
    assert np.abs(zero - (plain + dc)).max() < 1e-9
    # a constant phase only rotates the shifted part
    ph = O.rows_shifted(raw, nf, n, 0.0, 0.25, mode=O.MODE_COMPLEX)
    assert np.abs(ph - (1j * plain + dc)).max() < 1e-9


# ---------------------------------------------------------------------------------------------
# a5 against an FFTW3-API library, when the image has one (FFTW itself, or Intel MKL's libmkl_rt,
# which implements fftw_plan_dft_1d / fftw_execute): the calls the reference makes, src/nrf.c:562-615
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [128, 1024, 8192, 16384])
def test_oracle_fft_against_an_fftw3_api_library(golden, n):
    lib_name = O.find_fftw_api()
    if lib_name is None:
        pytest.skip("no FFTW3-API library in this image")
    raw = np.concatenate([golden["rf_100p900_1__flipped"], golden["rf_202p500_2__flipped"]]) ^ np.uint8(0x80)
    nf = raw.size // (2 * n)
    _, got = O.time_mag_rows_fftw(lib_name, raw, nf, n, threads=3, keep_rows=nf)
    want = O.rows(raw, nf, n)
    assert np.abs(got - want).max() <= 1e-9 * np.abs(want).max()
    if n in GOLDEN_SIZES:
        assert np.abs(got[0] - golden["rf_100p900_1__mag_%d" % n]).max() <= 1e-9 * np.abs(want).max()


def test_threaded_rows_are_the_single_threaded_rows():
    """orc_rows_mt (the full-size comparisons of the GPU tier) cuts the frames into contiguous ranges, one pthread each running
    the single-threaded function: the same rows bit for bit, every mode, with and without a taper, ragged thread counts."""
    n, nf, hop = 256, 37, 128
    iq = np.random.default_rng(9).integers(0, 256, 2 * ((nf - 1) * hop + n), dtype=np.uint8)
    w = O.window("blackman", n)
    for mode in (O.MODE_MAG, O.MODE_DB10_U8, O.MODE_DB5_U8_DCFIX, O.MODE_COMPLEX, O.MODE_MAG_NODC, O.MODE_DB_F64):
        for flip in (True, False):
            want = O.rows(iq, nf, n, hop=hop, flip=flip, mode=mode)
            for threads in (1, 3, 8, 64):
                assert np.array_equal(O.rows_mt(iq, nf, n, hop=hop, flip=flip, mode=mode, threads=threads), want), (mode, flip, threads)
        want = O.rows_windowed(iq, nf, n, w, hop=hop, mode=mode)
        assert np.array_equal(O.rows_mt(iq, nf, n, hop=hop, mode=mode, window=w, threads=5), want), mode
    assert O.rows_mt(iq, 0, n, hop=hop).shape == (0, n)
    assert 1 <= O.host_threads() <= 32


def test_windowed_shifted_and_f64_rows_reduce_to_their_unwindowed_forms():
    n, nf = 128, 6
    rng = np.random.default_rng(4)
    iq = rng.integers(0, 256, 2 * n * nf, dtype=np.uint8)
    ones = np.ones(n)
    for mode in (O.MODE_MAG, O.MODE_COMPLEX, O.MODE_DB5_U8_DCFIX):
        assert np.array_equal(O.rows_shifted_windowed(iq, nf, n, 0.013, ones, 0.4, mode=mode), O.rows_shifted(iq, nf, n, 0.013, 0.4, mode=mode))
    x = rng.normal(0.2, 0.3, 2 * n * nf)
    plain = np.stack([O.mag_row(O.fft_forward(O.unpack_center_f64(x[2 * f * n: 2 * (f + 1) * n]))) for f in range(nf)])
    assert np.array_equal(O.rows_f64(x, nf, n), plain) and np.array_equal(O.rows_f64(x, nf, n, window=ones), plain)
    # the taper sits beside the (-1)^n: against numpy
    w = O.window("hann", n)
    z = (x[0::2] + 1j * x[1::2]).reshape(nf, n) * ((-1.0) ** np.arange(n)) * w
    assert np.max(np.abs(O.rows_f64(x, nf, n, mode=O.MODE_COMPLEX, window=w) - np.fft.fft(z, axis=1))) < 1e-10


def _numpy_rows(raw, n_frames, n, hop, flip, mode):
    """An independent restatement in numpy (pocketfft, f64) of what O.rows computes: src/nrf.c:95-110 (flip), :599-614 (unpack +
    (-1)^n), :615 (forward DFT), :619-630 (magnitude + DC patch); c/fft-batch.c:83-94 and c/fft-batch-broad.c:106-121 (pixels)."""
    sign = 1.0 - 2.0 * (np.arange(n) & 1)
    out = []
    for f in range(n_frames):
        b = raw[2 * f * hop: 2 * (f * hop + n)]
        u = ((b.astype(np.int64) + 128) % 256 if flip else b.astype(np.int64)).astype(np.float64) / 256.0
        spec = np.fft.fft((u[0::2] + 1j * u[1::2]) * sign)
        if mode == O.MODE_MAG:
            row = np.abs(spec)
            row[n // 2] = row[n // 2 - 1]
        elif mode == O.MODE_MAG_NODC:
            row = np.abs(spec)
        elif mode == O.MODE_COMPLEX:
            row = spec
        else:
            scale = 10.0 if mode == O.MODE_DB10_U8 else 5.0
            d = 10.0 * np.log10(spec.real ** 2 + spec.imag ** 2 + 1e-20) * scale
            row = np.clip(np.trunc(d), 0, 255)
            if mode == O.MODE_DB5_U8_DCFIX:
                row[n // 2] = row[n // 2 - 1]
        out.append(row)
    return np.stack(out)


def test_rows_of_random_geometry_against_an_independent_numpy_restatement():
    """Beyond the fixed fixtures: sizes, frame counts, hops (overlapped, gapped), both byte conventions and every epilogue drawn
    at random (hypothesis), the oracle's C loop against numpy's pocketfft with the epilogues written out again -- frame
    indexing, flip, centring, DC patch, truncation and clamp.  Pixels may differ where the dB value sits within 1e-9 of an
    integer (two f64 FFTs round differently there); nowhere else."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=60, deadline=None, derandomize=True)
    @given(st.sampled_from([32, 64, 128, 256, 512, 1024, 2048, 4096]), st.integers(1, 9), st.sampled_from([0.5, 1.0, 1.25]),
           st.booleans(), st.sampled_from([O.MODE_MAG, O.MODE_MAG_NODC, O.MODE_COMPLEX, O.MODE_DB10_U8, O.MODE_DB5_U8_DCFIX]),
           st.integers(0, 2 ** 31 - 1))
    def check(n, n_frames, hop_frac, flip, mode, seed):
        hop = max(8, int(n * hop_frac) // 8 * 8)
        rng = np.random.default_rng(seed)
        raw = rng.integers(0, 256, 2 * ((n_frames - 1) * hop + n), dtype=np.uint8)
        if seed % 5 == 0:
            raw[:] = 0x80 if flip else 0x00                      # silence: log of (almost) nothing, the clamp at 0
        got = O.rows(raw, n_frames, n, hop=hop, flip=flip, mode=mode)
        want = _numpy_rows(raw, n_frames, n, hop, flip, mode)
        if mode in (O.MODE_DB10_U8, O.MODE_DB5_U8_DCFIX):
            diff = got.astype(np.int64) != want.astype(np.int64)
            if diff.any():                                       # only at a truncation boundary
                u = (raw.astype(np.int64) + 128) % 256 if flip else raw.astype(np.int64)
                assert np.count_nonzero(diff) <= 2 and np.abs(got.astype(np.int64) - want.astype(np.int64)).max() <= 1, (n, hop, mode, u[:4])
        else:
            scale = max(1.0, float(np.abs(want).max()))
            assert np.abs(got - want).max() <= 1e-11 * scale * np.log2(n), (n, n_frames, hop, flip, mode)

    check()
