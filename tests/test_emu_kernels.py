"""CPU tier: the kernel source of frequensea_amd/csrc/fsea_fft_core.h, compiled by g++ through
the test-only shim in tests/emu, against the oracle.  Catches index-arithmetic errors (Stockham
addressing, LDS padding, twiddle tables, epilogue bin ownership, frame mapping) without a GPU.
The GPU tier (test_gpu_parity.py) repeats these checks on the real device."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import parity
from tests.conftest import GOLDEN_KEYS, synth_iq
from tests.emu_util import emu_rows, emu_tiled

SIZES = [32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384]


def frames_for(n):
    return 70 if n <= 512 else (19 if n <= 2048 else 5)


@pytest.mark.parametrize("n", SIZES)
def test_mag_all_sizes(n):
    nf = frames_for(n)                      # not a multiple of frames-per-workgroup: ragged tail
    iq = synth_iq(n, 2 * nf * n)
    got = emu_rows(iq, n, nf, grid=2)
    parity.check_mode(got, iq, n, nf, n, True, 0)
    assert np.array_equal(got[:, n // 2], got[:, n // 2 - 1])      # DC patch


@pytest.mark.parametrize("n", [32, 256, 1024, 4096, 8192, 16384])
@pytest.mark.parametrize("mode", [1, 2])
def test_compile_time_pixel_kernels(n, mode):
    """`*_u8_db10` / `*_u8_db5`: epilogue and byte convention fixed at compile time (the sweep tools' path)."""
    nf = 9 if n <= 1024 else 3
    iq = synth_iq(300 + n + mode, 2 * nf * n)
    got = emu_rows(iq, n, nf, mode=mode, grid=2, specialised=True)
    parity.check_mode(got, iq, n, nf, n, True, mode)
    if mode == 2:
        assert np.array_equal(got[:, n // 2], got[:, n // 2 - 1])


@pytest.mark.parametrize("n", [32, 64, 128, 1024, 4096, 8192])
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4, 5])
def test_all_modes_runtime_dispatch(n, mode):
    nf = 9 if n <= 1024 else 3
    iq = synth_iq(100 + n + mode, 2 * nf * n)
    got = emu_rows(iq, n, nf, mode=mode, grid=1, specialised=False)
    parity.check_mode(got, iq, n, nf, n, True, mode)


@pytest.mark.parametrize("n,hop", [(256, 128), (1024, 512), (1024, 8), (8192, 4096), (16384, 8192)])
def test_overlapped_hop(n, hop):
    nf = 11 if n <= 1024 else 4
    iq = synth_iq(7 * n + hop, 2 * ((nf - 1) * hop + n))
    got = emu_rows(iq, n, nf, hop=hop, grid=3)
    parity.check_mode(got, iq, n, nf, hop, True, 0)


@pytest.mark.parametrize("n", [256, 2048, 8192])
def test_no_flip_rtlsdr_style(n):
    nf = 6
    iq = synth_iq(5 * n, 2 * nf * n) ^ np.uint8(0x80)     # already offset binary
    for mode in (0, 1):
        got = emu_rows(iq, n, nf, flip=False, mode=mode, specialised=(mode == 0))
        parity.check_mode(got, iq, n, nf, n, False, mode)


@pytest.mark.parametrize("grid", [1, 2, 8, 16])
def test_grid_mappings_cover_every_frame(grid):
    n, nf = 1024, 53
    iq = synth_iq(grid, 2 * nf * n)
    got = emu_rows(iq, n, nf, grid=grid)
    parity.check_mode(got, iq, n, nf, n, True, 0)


@pytest.mark.parametrize("n", [128, 1024, 4096])
def test_f32_complex_input(n):
    """The NUT_BUFFER_F64 branch of nrf_fft_process (src/nrf.c:607-609) arrives as f32 pairs."""
    nf = 5
    rng = np.random.default_rng(n)
    x = rng.normal(0, 0.2, 2 * nf * n)
    got = emu_rows(x.astype(np.float32), n, nf, mode=0, in_kind=1, specialised=False)
    spec = np.stack([O.fft_forward(O.unpack_center_f64(x[2 * f * n: 2 * (f + 1) * n])) for f in range(nf)])
    want = np.stack([O.mag_row(s) for s in spec])
    parity.check_float(got, want)
    gotc = emu_rows(x.astype(np.float32), n, nf, mode=3, in_kind=1, specialised=False)
    parity.check_float(gotc, spec)


@pytest.mark.parametrize("key", GOLDEN_KEYS)
@pytest.mark.parametrize("n", [256, 1024, 8192])
def test_recorded_captures(golden, key, n):
    raw = golden[key + "__raw"]
    got = emu_rows(raw, n, 1, grid=1)
    parity.check_float(got[0], golden["%s__mag_%d" % (key, n)])
    px = emu_rows(raw, n, 1, mode=1, grid=1, specialised=False)
    parity.check_u8(px[0], golden["%s__db10_%d" % (key, n)])
    px = emu_rows(raw, n, 1, mode=2, grid=1, specialised=False)
    parity.check_u8(px[0], golden["%s__db5_%d" % (key, n)])


def test_edge_inputs():
    n = 1024
    # constant input: only the (patched-away) DC bin is non-zero
    z = np.zeros(2 * n, np.uint8)
    assert emu_rows(z, n, 1).max() < 1e-4
    spec = emu_rows(z, n, 1, mode=3, specialised=False)[0]
    assert abs(abs(spec[n // 2]) - 0.5 * n * np.sqrt(2)) < 1e-3 and np.abs(np.delete(spec, n // 2)).max() < 1e-4
    # full-scale extremes
    ext = np.tile(np.array([0x7f, 0x80, 0x80, 0x7f], np.uint8), n // 2)
    got = emu_rows(ext, n, 1)
    parity.check_mode(got, ext, n, 1, n, True, 0)
    # zero frames: nothing is written
    assert emu_rows(z, n, 0).shape == (0, n)


@pytest.mark.parametrize("n,variant", [(8192, "nd"), (8192, "v2"), (8192, "v2s"),
                                       (8192, "x0"), (8192, "A"), (8192, "B"), (8192, "D"), (8192, "B2"), (8192, "D2"), (8192, "W"),
                                       (8192, "notwl"), (8192, "notwr"), (4096, "nr"), (2048, "nr"), (4096, "x0"), (4096, "df"), (4096, "t256"), (4096, "B3"), (4096, "B"),
                                       (4096, "C"), (4096, "D"), (2048, "x0"), (2048, "df"), (2048, "B"), (2048, "C"),
                                       (1024, "x0"), (1024, "B"), (1024, "C"), (1024, "D"),
                                       (16384, "nd"), (16384, "B"), (256, "p16"), (128, "p16"),
                                       (4096, "w64"), (4096, "w64b"), (4096, "s2"), (4096, "pk"), (4096, "px0"), (8192, "pk"), (8192, "px0"),
                                       (256, "pk"), (256, "px0"), (1024, "px0"), (256, "p64"), (1024, "r2"), (1024, "e"), (1024, "h")])
def test_tuning_variants(n, variant):
    """Every kernel variant compiled into libfsea_hip_tune.so (fsea_plan_create_variant) stays correct."""
    nf = 9 if n <= 1024 else 3
    iq = synth_iq(n + len(variant), 2 * nf * n)
    for mode in (0, 1, 3):
        got = emu_rows(iq, n, nf, mode=mode, grid=2, specialised=(mode == 0), variant=variant)
        parity.check_mode(got, iq, n, nf, n, True, mode)


def test_random_geometry_sweep():
    """Seeded random draws of (size, hop, frame count, mode, flip, grid): overlapped, gapped
    (hop > N) and tiny hops, ragged frame counts, every epilogue."""
    rng = np.random.default_rng(2026)
    for _ in range(28):
        n = int(rng.choice([32, 64, 128, 256, 512, 1024, 2048, 4096, 8192]))
        hop = int(rng.choice([8, 16, n // 4, n // 2, n, n + 8, 2 * n]))
        nf = int(rng.integers(1, 40 if n <= 1024 else 7))
        mode = int(rng.integers(0, 6))
        flip = bool(rng.integers(0, 2))
        grid = int(rng.choice([1, 2, 3, 8]))
        iq = synth_iq(int(rng.integers(1 << 30)), 2 * ((nf - 1) * hop + n))
        got = emu_rows(iq, n, nf, hop=hop, flip=flip, mode=mode, grid=grid, specialised=bool(rng.integers(0, 2)))
        try:
            parity.check_mode(got, iq, n, nf, hop, flip, mode)
        except AssertionError as e:
            raise AssertionError("n=%d hop=%d nf=%d mode=%d flip=%s grid=%d: %s" % (n, hop, nf, mode, flip, grid, e))


@pytest.mark.parametrize("n", [128, 256, 512, 1024, 2048, 4096, 8192, 16384])
def test_frequency_shifted_input(n):
    """The ROT kernels (fsea_exec_u8_shifted_*): nrf_freq_shifter fused into the load, against
    the oracle's shifter + F64-branch FFT.  150 kHz at 5 Msps as lua/fft-shifted.lua would ask."""
    nf = 5 if n <= 2048 else 3
    iq = synth_iq(500 + n, 2 * nf * n)
    delta, phase0 = 150000 / 5000000, 0.3125
    for mode in (0, 3, 1):
        got = emu_rows(iq, n, nf, mode=mode, grid=2, shift=(delta, phase0))
        parity.check_mode_shifted(got, iq, n, nf, n, True, mode, delta, phase0)


def test_frequency_shift_overlap_negative_offset_and_no_flip():
    n, hop, nf = 1024, 256, 9
    iq = synth_iq(77, 2 * ((nf - 1) * hop + n))
    for delta, flip in ((-0.123456789, True), (0.49, False), (0.0, True), (7.25, True)):
        got = emu_rows(iq, n, nf, hop=hop, flip=flip, mode=3, grid=3, shift=(delta, 0.0))
        parity.check_mode_shifted(got, iq, n, nf, hop, flip, 3, delta, 0.0)


@pytest.mark.parametrize("n,tile_rows,mode", [(8192, 3, 2), (4096, 4, 2), (4096, 2, 0), (1024, 8, 1), (256, 32, 2), (64, 64, 0)])
def test_tiled_output_addressing(n, tile_rows, mode):
    """fsea_exec_u8_tiled_device's addressing in the kernel source: tile k of the launch lands at columns
    first_x + k * tile_step of rows 0..tile_rows-1, everything else in the image stays as it was."""
    tiles, first_x, step = 3, 8, n + 12
    nf = tiles * tile_rows
    iq = synth_iq(n + tile_rows, 2 * nf * n)
    rows = emu_rows(iq, n, nf, mode=mode, grid=2)
    fill = 7
    shape = (tile_rows + 1, first_x + (tiles - 1) * step + n + 4)
    image = emu_tiled(iq, n, nf, shape, first_x, tile_rows, step, mode=mode, grid=2, fill=fill)
    want = np.full(shape, fill, dtype=rows.dtype)
    for k in range(tiles):
        want[:tile_rows, first_x + k * step: first_x + k * step + n] = rows[k * tile_rows:(k + 1) * tile_rows]
    assert np.array_equal(image, want)


@pytest.mark.parametrize("n,nf,grid", [(8192, 7, 3), (8192, 2, 4), (4096, 9, 2), (4096, 3, 3), (16384, 5, 2)])
def test_static_unit_interleave_of_the_multi_wave_sizes(n, nf, grid):
    """Short launches of the 4096 / 8192 / 16384-point kernels run without the ticket pools (FftArgs::dynamic_units = 0):
    ragged unit counts, more workgroups than units, odd frame counts at two frames per workgroup."""
    iq = synth_iq(n + nf, 2 * nf * n)
    for mode in (0, 2):
        got = emu_rows(iq, n, nf, mode=mode, grid=grid, dynamic_units=False)
        parity.check_mode(got, iq, n, nf, n, True, mode)


@pytest.mark.parametrize("variant", ["w64", "w64b"])
def test_single_wave_64x64_schedule(variant):
    """FftKernel::run_w64 (tuning library): dword loads redistributed by v_permlane32_swap, one LDS exchange, deferred
    last-pass twiddles, pixel rows transposed in quads -- compile-time and run-time-mode kernels, both byte conventions,
    ragged grids, and the tiled (stitched-image) addressing."""
    n = 4096
    for nf, grid in ((1, 1), (5, 3), (4, 8)):
        iq = synth_iq(64 + nf, 2 * nf * n)
        for mode in (0, 1, 2):
            for spec in (True, False):
                got = emu_rows(iq, n, nf, mode=mode, grid=grid, specialised=spec, variant=variant)
                parity.check_mode(got, iq, n, nf, n, True, mode)
                if mode == 2:
                    assert np.array_equal(got[:, n // 2], got[:, n // 2 - 1])
    iq = synth_iq(65, 2 * 3 * n) ^ np.uint8(0x80)
    for mode in (0, 2, 3):
        got = emu_rows(iq, n, 3, flip=False, mode=mode, specialised=False, variant=variant)
        parity.check_mode(got, iq, n, 3, n, False, mode)
    # overlapped frames (hop < N) and the tiled output
    hop, nf = 1024, 6
    iq = synth_iq(66, 2 * ((nf - 1) * hop + n))
    got = emu_rows(iq, n, nf, hop=hop, mode=1, grid=2, variant=variant)
    parity.check_mode(got, iq, n, nf, hop, True, 1)
    tiles, tile_rows, first_x, step = 2, 3, 8, n + 12
    nf = tiles * tile_rows
    iq = synth_iq(67, 2 * nf * n)
    rows = emu_rows(iq, n, nf, mode=2, grid=2, variant=variant)
    shape = (tile_rows + 1, first_x + (tiles - 1) * step + n + 4)
    image = emu_tiled(iq, n, nf, shape, first_x, tile_rows, step, mode=2, grid=2, fill=7, variant=variant)
    want = np.full(shape, 7, dtype=np.uint8)
    for k in range(tiles):
        want[:tile_rows, first_x + k * step: first_x + k * step + n] = rows[k * tile_rows:(k + 1) * tile_rows]
    assert np.array_equal(image, want)


def test_biased_pixel_rounding_equals_truncation_on_a_dense_sweep():
    """The product's pixel epilogue lets v_cvt_pk_u8_f32 round to nearest on d - (0.5 - 2^-25) instead of truncating d
    (fsea_fft_core.h, OPT 4194304).  On a dense sweep of d over the pixel range the two differ only where d lies within
    ~1e-5 above an odd integer."""
    d = np.linspace(-3.0, 260.0, 4_000_001).astype(np.float32)
    biased = np.clip(np.rint((d.astype(np.float64) - 0.49999997).astype(np.float32)), 0, 255)
    trunc = np.clip(np.trunc(d), 0, 255)
    diff = biased != trunc
    assert diff.mean() < 1e-4
    assert np.all(np.abs(biased - trunc)[diff] == 1)
    frac = d[diff] - np.floor(d[diff])
    assert np.all(frac < 2e-5)


@pytest.mark.parametrize("n,nf,grid,run_len", [(8192, 9, 2, 4), (8192, 5, 3, 1), (8192, 7, 2, 8), (16384, 6, 2, 3), (16384, 4, 1, 8)])
def test_half_overlap_runs_keep_half_a_frame_in_registers(n, nf, grid, run_len):
    """FftKernel<..., RUNS = true> (K_U8_MAG_HALF, hop == N/2): a workgroup takes runs of consecutive frames and keeps
    pass-0 rows R0/2 .. R0-1 of a frame as rows 0 .. R0/2-1 of the next; ragged last runs, more workgroups than runs,
    run length 1 (no reuse at all) -- the rows are those of the ordinary kernel, bit for bit."""
    hop = n // 2
    iq = synth_iq(n + nf + run_len, 2 * ((nf - 1) * hop + n))
    got = emu_rows(iq, n, nf, hop=hop, grid=grid, run_len=run_len)
    parity.check_mode(got, iq, n, nf, hop, True, 0)
    plain = emu_rows(iq, n, nf, hop=hop, grid=grid, dynamic_units=False)
    assert np.array_equal(got, plain)


# ---- taper window (FftKernel<..., WIN>; fsea_plan_set_window) ----

def _taper(name, n):
    if name == "random":
        return np.random.default_rng(n).uniform(-1.0, 2.0, n).astype(np.float32)
    return O.window(name, n).astype(np.float32)


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("wname,wmode", [("hann", 2), ("random", 1)])
def test_windowed_kernels_all_sizes(n, wname, wmode):
    """The windowed MAG kernel (weights register-resident / fetched per frame; centred form for the cosine-sum taper,
    offset-binary form for the arbitrary one) against the oracle's windowed rows."""
    nf = 21 if n <= 512 else (9 if n <= 2048 else 3)
    iq = synth_iq(11 * n, 2 * nf * n)
    w = _taper(wname, n)
    got = emu_rows(iq, n, nf, window=w, window_mode=wmode, grid=2)
    parity.check_mode_windowed(got, iq, n, nf, n, True, 0, w)


@pytest.mark.parametrize("n", [32, 128, 1024, 4096])
@pytest.mark.parametrize("mode", [1, 2, 3, 4, 5])
def test_windowed_run_time_mode_kernel(n, mode):
    nf = 7 if n <= 1024 else 3
    iq = synth_iq(400 + n + mode, 2 * nf * n)
    for wname, flip, form in (("blackman", True, 0), ("random", False, 0), ("hann", False, 2)):
        w = _taper(wname, n)
        got = emu_rows(iq, n, nf, flip=flip, mode=mode, window=w, window_mode=1 + (mode & 1), window_form=form)
        parity.check_mode_windowed(got, iq, n, nf, n, flip, mode, w)
        if mode in (1, 2) and flip:     # the compile-time windowed pixel kernel and the run-time-mode one: the same pixels
            rt = emu_rows(iq, n, nf, flip=flip, mode=mode, window=w, window_mode=1 + (mode & 1), window_form=form, specialised=False)
            assert np.array_equal(got, rt)


@pytest.mark.parametrize("n", [64, 512, 2048, 8192])
def test_unit_window_gives_the_unwindowed_kernels_bits(n):
    nf = 5 if n <= 2048 else 2
    iq = synth_iq(77 + n, 2 * nf * n)
    ones = np.ones(n, np.float32)
    for mode, spec in ((0, True), (1, False), (3, False)):
        base = emu_rows(iq, n, nf, mode=mode, specialised=spec)
        for wmode in (1, 2):
            got = emu_rows(iq, n, nf, mode=mode, specialised=spec, window=ones, window_mode=wmode)
            assert np.array_equal(base.view(np.uint8), got.view(np.uint8)), (n, mode, wmode)


@pytest.mark.parametrize("n", [8192, 16384])
def test_windowed_half_overlap_kernel(n):
    hop, nf = n // 2, 7
    iq = synth_iq(n + 5, 2 * ((nf - 1) * hop + n))
    w = _taper("hann", n)
    got = emu_rows(iq, n, nf, hop=hop, run_len=3, window=w, window_mode=2, grid=2)
    parity.check_mode_windowed(got, iq, n, nf, hop, True, 0, w)


@pytest.mark.parametrize("n", [32, 256, 1024, 4096, 8192])
def test_windowed_shifted_and_f32_input_kernels(n):
    """K_U8_ROT_WIN and K_F32_WIN: the frequency-shifted and the f32-complex-input kernels with the taper (both forms of the
    DC term for the shifted one) against the oracle."""
    nf = 3
    iq = synth_iq(900 + n, 2 * nf * n)
    for wname, form in (("hann", 0), ("hann", 2)):
        w = _taper(wname, n)
        for flip in (True, False):
            got = emu_rows(iq, n, nf, flip=flip, mode=3, shift=(0.0137, 0.25), window=w, window_form=form, specialised=False)
            want = O.rows_shifted_windowed(iq, nf, n, 0.0137, w.astype(np.float64), 0.25, flip=flip, mode=O.MODE_COMPLEX)
            parity.check_float(got, want)
    x = np.random.default_rng(n).normal(0.2, 0.3, 2 * nf * n).astype(np.float32)
    w = _taper("blackman", n)
    got = emu_rows(x, n, nf, mode=0, in_kind=1, specialised=False, window=w)
    parity.check_float(got, O.rows_f64(x.astype(np.float64), nf, n, window=w.astype(np.float64)))
    ones = np.ones(n, np.float32)
    assert np.array_equal(emu_rows(x, n, nf, mode=3, in_kind=1, specialised=False, window=ones),
                          emu_rows(x, n, nf, mode=3, in_kind=1, specialised=False))


def test_windowed_random_geometry_sweep():
    """Seeded random draws of (size, hop, frame count, mode, flip, grid, taper, weight residency) through the windowed
    kernels: overlapped, gapped and tiny hops, ragged frame counts, every epilogue, cosine-sum / arbitrary / signed weights."""
    rng = np.random.default_rng(4026)
    for _ in range(24):
        n = int(rng.choice([32, 64, 128, 256, 512, 1024, 2048, 4096]))
        hop = int(rng.choice([8, 16, n // 4, n // 2, n, n + 8, 2 * n]))
        nf = int(rng.integers(1, 30 if n <= 1024 else 6))
        mode = int(rng.integers(0, 6))
        flip = bool(rng.integers(0, 2))
        grid = int(rng.choice([1, 2, 3, 8]))
        kind = str(rng.choice(["hann", "hamming", "flattop", "random", "ramp"]))
        w = (np.linspace(-1.0, 1.0, n).astype(np.float32) if kind == "ramp" else _taper(kind, n))
        iq = synth_iq(int(rng.integers(1 << 30)), 2 * ((nf - 1) * hop + n))
        got = emu_rows(iq, n, nf, hop=hop, flip=flip, mode=mode, grid=grid, specialised=bool(rng.integers(0, 2)), window=w,
                       window_mode=int(rng.integers(1, 3)))
        try:
            parity.check_mode_windowed(got, iq, n, nf, hop, flip, mode, w)
        except AssertionError as e:
            raise AssertionError("n=%d hop=%d nf=%d mode=%d flip=%s grid=%d taper=%s: %s" % (n, hop, nf, mode, flip, grid, kind, e))


def test_window_tables_decide_the_form():
    """build_window_tables (fsea_tables.h, through the emulated launch): cosine-sum tapers qualify for the centred form --
    then forcing the offset-binary form gives the same rows within tolerance, not the same bits."""
    n, nf = 256, 4
    iq = synth_iq(9, 2 * nf * n)
    w = _taper("flattop", n)
    a = emu_rows(iq, n, nf, mode=3, window=w)
    b = emu_rows(iq, n, nf, mode=3, window=w, window_form=2)
    assert not np.array_equal(a, b)
    parity.check_mode_windowed(a, iq, n, nf, n, True, 3, w)
    parity.check_mode_windowed(b, iq, n, nf, n, True, 3, w)


# ---- per-(size, mode) product configurations (fsea_configs.h: FSEA_CFG_256_ROWS, FSEA_CFG_512_PX, FSEA_CFG_1024_RT) ----

@pytest.mark.parametrize("n,variant", [(256, "rows"), (512, "px"), (1024, "rt")])
def test_per_mode_product_configurations(n, variant):
    """Full kernel sets: every mode, both byte conventions, f32 input, the frequency shift, a window, tiles."""
    nf = 37
    iq = synth_iq(900 + n, 2 * nf * n)
    for mode in range(6):
        for spec in (True, False):
            got = emu_rows(iq, n, nf, mode=mode, specialised=spec, variant=variant)
            parity.check_mode(got, iq, n, nf, n, True, mode)
    off = iq ^ np.uint8(0x80)
    parity.check_mode(emu_rows(off, n, nf, flip=False, mode=3, variant=variant), off, n, nf, n, False, 3)
    got = emu_rows(iq, n, nf, mode=0, variant=variant, shift=(0.013, 0.2))
    parity.check_mode_shifted(got, iq, n, nf, n, True, 0, 0.013, 0.2)
    w = _taper("hann", n)
    for mode in (0, 2, 3):
        got = emu_rows(iq, n, nf, mode=mode, variant=variant, window=w)
        parity.check_mode_windowed(got, iq, n, nf, n, True, mode, w)
    hop = n // 2
    got = emu_rows(iq, n, 2 * nf - 1, hop=hop, variant=variant)
    parity.check_mode(got, iq, n, 2 * nf - 1, hop, True, 0)
