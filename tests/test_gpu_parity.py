"""GPU tier (-m gpu): the HIP path, called through the C ABI of libfsea_hip.so and through the
reference-shaped nrf_fft API of libfsea_nrf.so, against the f64 oracle, the committed golden
vectors from the reference's recorded captures, and -- at BASELINE.json's full sizes -- through
size-independent properties (Parseval, tone position) and every row against the threaded oracle.  Tolerances: tests/parity.py."""
import ctypes
import os

import numpy as np
import pytest

from frequensea_amd import fsea, nrf
from oracle import oracle as O
from tests import parity
from tests.conftest import ALL_CAPTURE_KEYS, GOLDEN_KEYS, GOLDEN_SIZES, ROOT, kernel_stem, synth_iq

pytestmark = pytest.mark.gpu

SIZES = [32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384]


class DeviceBuffer:
    def __init__(self, nbytes, device=0):
        self.ptr = ctypes.c_void_p()
        self.nbytes = nbytes
        self.device = device
        fsea._check(fsea.hip_lib().fsea_device_alloc(device, nbytes, ctypes.byref(self.ptr)))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        fsea._check(fsea.hip_lib().fsea_copy_to_device(self.device, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        fsea._check(fsea.hip_lib().fsea_copy_to_host(self.device, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            fsea.hip_lib().fsea_device_free(self.device, self.ptr)
            self.ptr = ctypes.c_void_p()


@pytest.fixture(params=["auto", "tickets", "static"])
def units_policy(request, monkeypatch):
    """Runs a test with the frame distribution of the multi-wave sizes left to the library (per launch length) and
    pinned either way (fsea_plan_set_unit_distribution): short test launches would otherwise never reach the ticket pools."""
    policy = {"auto": fsea.UNITS_AUTO, "tickets": fsea.UNITS_TICKETS, "static": fsea.UNITS_STATIC}[request.param]
    real = fsea.Plan

    def make(*args, **kwargs):
        plan = real(*args, **kwargs)
        plan.set_unit_distribution(policy)
        return plan

    monkeypatch.setattr(fsea, "Plan", make)
    return request.param


def test_device_present_and_library_loaded():
    assert fsea.device_count() >= 1
    p = fsea.Plan(8192)
    assert p.kernel_name == "fsea_fft8192_u8_mag"
    grid, block, lds = p.grid(4096)
    assert block == 256 and grid >= 256 and grid % 8 == 0 and lds > 64 * 1024
    p.close()


def test_committed_traffic_figure_belongs_to_the_current_kernel():
    """bench.py quotes roofline.traffic from profiles/traffic.json, a PMC measurement made once per round.
    It goes stale when the default kernel changes: kernel name, grid, workgroup size and LDS bytes of the
    headline plan must still be the ones it was measured on."""
    import json
    tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    p = fsea.Plan(8192)
    assert tr["kernel"] == p.kernel_name
    assert tr["grid_block_lds"] == list(p.grid(tr["frames"])), "re-run scripts/pmc.sh and update profiles/traffic.json"
    p.close()


@pytest.mark.parametrize("n", SIZES)
def test_mag_rows_all_sizes(n):
    nf = 300 if n <= 1024 else 37
    iq = synth_iq(n, 2 * nf * n)
    plan = fsea.Plan(n)
    got = plan.exec_host(iq, nf)
    parity.check_mode(got, iq, n, nf, n, True, 0)
    assert np.array_equal(got[:, n // 2], got[:, n // 2 - 1])
    plan.close()


@pytest.mark.parametrize("n", [32, 64, 128, 1024, 4096, 8192, 16384])
@pytest.mark.parametrize("mode", [1, 2, 3, 4, 5])
def test_other_modes(n, mode):
    nf = 33
    iq = synth_iq(1000 + n + mode, 2 * nf * n)
    plan = fsea.Plan(n, mode=mode)
    got = plan.exec_host(iq, nf)
    parity.check_mode(got, iq, n, nf, n, True, mode)
    plan.close()


@pytest.mark.parametrize("n,hop", [(256, 128), (1024, 8), (4096, 2048), (16384, 8192)])
def test_overlapped_frames(n, hop):
    nf = 41
    iq = synth_iq(3 * n + hop, 2 * ((nf - 1) * hop + n))
    plan = fsea.Plan(n, hop=hop)
    got = plan.exec_host(iq, nf)
    parity.check_mode(got, iq, n, nf, hop, True, 0)
    plan.close()


def test_random_geometry_sweep_on_gpu(units_policy):
    rng = np.random.default_rng(7)
    for _ in range(40):
        n = int(rng.choice(SIZES))
        hop = int(rng.choice([8, 16, n // 4, n // 2, n, n + 8, 2 * n]))
        nf = int(rng.integers(1, 3000 if n <= 1024 else 300))
        mode = int(rng.integers(0, 6))
        flip = bool(rng.integers(0, 2))
        iq = synth_iq(int(rng.integers(1 << 30)), 2 * ((nf - 1) * hop + n))
        plan = fsea.Plan(n, hop=hop, mode=mode)
        got = plan.exec_host(iq, nf, flip=flip)
        k = min(nf, 24)                                   # oracle time: first and last rows
        rows = np.r_[0:k // 2, nf - (k - k // 2):nf]
        for f in np.unique(rows):
            sub = iq[2 * f * hop: 2 * (f * hop + n)]
            try:
                parity.check_mode(got[f:f + 1], sub, n, 1, n, flip, mode)
            except AssertionError as e:
                raise AssertionError("n=%d hop=%d nf=%d mode=%d flip=%s row %d: %s" % (n, hop, nf, mode, flip, f, e))
        plan.close()


def test_flip_matches_the_reference_byte_flip():
    """a1 (src/nrf.c:100-109): (b + 128) % 256 on every byte.  The oracle's flip is that loop; the kernel
    folds it into the conversion.  All 256 byte values in both components, against the oracle with
    flip (raw int8) and without (bytes flipped by the oracle first), and the two GPU paths bit for bit."""
    n, nf = 1024, 64
    raw = synth_iq(5, 2 * nf * n)
    raw[:512] = np.repeat(np.arange(256, dtype=np.uint8), 2)             # every byte value as I and as Q
    flipped = O.flip_u8(raw)
    assert np.array_equal(flipped, ((raw.astype(np.int32) + 128) % 256).astype(np.uint8))
    plan = fsea.Plan(n, mode=fsea.MODE_COMPLEX_F32)
    a = plan.exec_host(raw, nf, flip=True)
    b = plan.exec_host(flipped, nf, flip=False)
    parity.check_mode(a, raw, n, nf, n, True, 3)
    parity.check_mode(b, flipped, n, nf, n, False, 3)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    plan.close()


@pytest.mark.parametrize("key", GOLDEN_KEYS)
@pytest.mark.parametrize("n", GOLDEN_SIZES)
def test_recorded_captures_match_golden(golden, key, n):
    raw = golden[key + "__raw"]
    for mode, name in [(0, "mag"), (1, "db10"), (2, "db5")]:
        plan = fsea.Plan(n, mode=mode)
        got = plan.exec_host(raw, 1)[0]
        want = golden["%s__%s_%d" % (key, name, n)]
        if mode == 0:
            parity.check_float(got, want)
        else:
            parity.check_u8(got, want)
        plan.close()


@pytest.mark.parametrize("key", ALL_CAPTURE_KEYS)
def test_every_recorded_capture_matches_golden(golden_all, key):
    """BASELINE.json: "outputs match FFTW on the same rfdata/*.raw inputs" -- all 35 full-size captures of the reference's
    rfdata/ (tests/golden/rfdata_all_golden.npz: scipy/pocketfft f64 rows), at the nrf_* size and the headline size."""
    raw = golden_all[key + "__raw"]
    for n in (1024, 8192):
        plan = fsea.Plan(n)
        got = plan.exec_host(raw, 1)[0]
        parity.check_float(got, golden_all["%s__mag_%d" % (key, n)])
        plan.close()


def test_whole_block_of_one_capture_through_the_batched_entry_and_the_nrf_api(golden_all, tmp_path):
    """One 262144-byte block of a recorded capture (one libhackrf transfer): its 128 consecutive 1024-point frames in one
    launch, and the same block through nrf_device_new (file replay) -> nrf_device_step -> nrf_fft_process ->
    nrf_fft_get_buffer, which transforms the block's first 1024 samples per step (src/nrf.c:153-170, 598-631)."""
    raw = golden_all["block__raw"]
    want = golden_all["block__mag_1024"]
    plan = fsea.Plan(1024)
    got = plan.exec_host(raw, 128)
    parity.check_float(got, want)
    plan.close()
    path = tmp_path / "block.raw"
    raw.tofile(path)
    L = nrf.nrf_lib()
    dev = L.nrf_device_new(100.0, str(path).encode())   # no radio on the box: the file-replay device (src/nrf.c:309-318)
    L.nrf_device_set_paused(dev, 1)
    fft = L.nrf_fft_new(1024, 4)
    import time
    for _ in range(4):
        L.nrf_device_step(dev)
        time.sleep(0.06)                                    # > 2 replay periods of 1/60 s (the receive thread's pace)
        buf = L.nrf_device_get_samples_buffer(dev)
        assert np.array_equal(nrf.buffer_to_numpy(L, buf), O.flip_u8(raw))   # the byte flip of src/nrf.c:100-109
        L.nrf_fft_process(fft, buf)
        L.nut_buffer_free(buf)
    out = L.nrf_fft_get_buffer(fft)
    hist = nrf.buffer_to_numpy(L, out).reshape(4, 1024)
    L.nut_buffer_free(out)
    for r in range(4):
        parity.check_float(hist[r], want[0])
    L.nrf_fft_free(fft)
    L.nrf_device_free(dev)


def test_survey_known_answers_on_gpu(golden):
    raw = golden["rf_100p900_1__raw"]
    row = fsea.Plan(1024).exec_host(raw, 1)[0]
    assert row.argmax() == 513 and abs(float(row.max()) - 37.935861) < 1e-4
    assert abs(float(row.astype(np.float64).sum()) - 1567.312172) < 1e-2
    row = fsea.Plan(8192).exec_host(raw, 1)[0]
    assert row.argmax() == 3358 and abs(float(row.max()) - 162.437851) < 5e-4


@pytest.mark.parametrize("n,mode", [(256, 0), (256, 4), (256, 5), (512, 1), (512, 2), (1024, 3), (1024, 4), (1024, 5)])
def test_per_mode_configurations(n, mode, monkeypatch):
    """Where the modes of one size prefer different radix orders a plan takes its mode's configuration (256 points: f32
    rows; 512: u8 pixels; 1024: the run-time-mode kernel's modes -- full kernel sets): against the oracle through every entry point, and
    against the size's first configuration (FSEA_ONE_CONFIG_PER_SIZE=1 at plan creation)."""
    nf = 301
    iq = synth_iq(60 + n + mode, 2 * nf * n)
    plan = fsea.Plan(n, mode=mode)
    suffix = {0: "_u8_mag", 1: "_u8_db10", 2: "_u8_db5"}.get(mode, "_u8")
    assert plan.kernel_name == kernel_stem(n, mode) + suffix and kernel_stem(n, mode) != "fsea_fft%d" % n
    got = plan.exec_host(iq, nf)
    parity.check_mode(got, iq, n, nf, n, True, mode)
    off = iq ^ np.uint8(0x80)
    got_off = plan.exec_host(off, nf, flip=False)                        # the run-time-mode kernel of the same configuration
    if mode in (1, 2):
        assert np.array_equal(got, got_off)
    parity.check_mode(got_off, off, n, nf, n, False, mode)
    parity.check_mode_shifted(plan.exec_shifted_host(iq, nf, 0.0173, 0.25), iq, n, nf, n, True, mode, 0.0173, 0.25)
    x = (off.astype(np.float64) / 256.0)
    got64 = plan.exec_host_f64(x, nf)                                    # f64 input = u8 / 256 (src/nrf.c:607-609)
    parity.check_mode(got64, off, n, nf, n, False, mode)
    plan.close()
    monkeypatch.setenv("FSEA_ONE_CONFIG_PER_SIZE", "1")
    first = fsea.Plan(n, mode=mode)
    assert first.kernel_name == "fsea_fft%d%s" % (n, suffix)
    base = first.exec_host(iq, nf)
    first.close()
    if mode in (1, 2):
        parity.check_u8(got, base)
    elif mode != 5:                                                       # (dB of numerically empty bins: oracle check above)
        assert np.linalg.norm(got.astype(np.complex128) - base) <= 1e-6 * np.linalg.norm(base)


@pytest.mark.parametrize("n", [32, 256, 1024, 4096, 8192, 16384])
def test_compile_time_pixel_kernels(n):
    """The sweep tools' modes on raw int8 input run `*_u8_db5` / `*_u8_db10` (epilogue and byte
    convention fixed at compile time); offset-binary input of the same plan runs `*_u8`.  Both
    against the oracle's pixel rules (c/fft-batch.c:83-94, c/fft-batch-broad.c:106-121)."""
    nf = 257 if n <= 1024 else 29
    iq = synth_iq(40 + n, 2 * nf * n)
    for mode, suffix in ((fsea.MODE_DB5_U8_DCFIX, "_u8_db5"), (fsea.MODE_DB10_U8, "_u8_db10")):
        plan = fsea.Plan(n, mode=mode)
        assert plan.kernel_name == "fsea_fft%d%s" % (n, suffix)
        got = plan.exec_host(iq, nf, flip=True)
        parity.check_mode(got, iq, n, nf, n, True, mode)
        got2 = plan.exec_host(iq ^ np.uint8(0x80), nf, flip=False)      # run-time-mode kernel, same pixels
        assert np.array_equal(got, got2)
        plan.close()


@pytest.mark.parametrize("n,tile_rows,mode", [(8192, 3, 2), (8192, 5, 0), (4096, 6, 2), (4096, 256, 1), (2048, 8, 3), (1024, 16, 2),
                                              (256, 32, 2), (64, 64, 0), (16384, 2, 2)])
def test_tiled_exec_writes_the_stitched_image_in_place(n, tile_rows, mode):
    """fsea_exec_u8_tiled_device: the rows of the launch land as side-by-side tiles of an image and equal, bit for
    bit, the plain rows of the same plan max-composited by the oracle (c/fft-stitch-broad.c:62-87) onto a zeroed
    image; bytes outside the tiles (gaps when tile_step > N, the rows below, the margin left of first_x) keep
    what they held."""
    tiles, first_x, step = 5, 16, n + 24
    nf = tiles * tile_rows
    iq = synth_iq(n + tile_rows, 2 * nf * n)
    dt = {0: np.float32, 1: np.uint8, 2: np.uint8, 3: np.complex64}[mode]
    plan = fsea.Plan(n, mode=mode)
    d_in = DeviceBuffer(iq.nbytes).upload(iq)
    d_rows = DeviceBuffer(nf * n * np.dtype(dt).itemsize)
    plan.exec_device(d_in.ptr, nf, d_rows.ptr)
    plan.synchronize()
    rows = d_rows.download(dt, (nf, n))
    shape = (tile_rows + 2, first_x + (tiles - 1) * step + n + 8)
    fill = np.full(shape, 3, dtype=dt)
    d_img = DeviceBuffer(fill.nbytes).upload(fill)
    plan.exec_tiled_device(d_in.ptr, nf, d_img.ptr, shape[0], shape[1], first_x, tile_rows, step)
    plan.synchronize()
    got = d_img.download(dt, shape)
    want = fill.copy()
    for k in range(tiles):
        want[:tile_rows, first_x + k * step: first_x + k * step + n] = rows[k * tile_rows:(k + 1) * tile_rows]
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
    if mode in (1, 2):                                          # the reference's operation on a zeroed image
        zero = np.zeros(shape, np.uint8)
        d_img.upload(zero)
        plan.exec_tiled_device(d_in.ptr, nf, d_img.ptr, shape[0], shape[1], first_x, tile_rows, step)
        plan.synchronize()
        ref = zero.copy()
        for k in range(tiles):
            O.composite_max(ref[:tile_rows], np.ascontiguousarray(rows[k * tile_rows:(k + 1) * tile_rows]), first_x + k * step)
        assert np.array_equal(d_img.download(np.uint8, shape), ref)
        parity.check_mode(rows, iq, n, nf, n, True, mode)
    # argument checks: ragged tiles, overlapping tiles, a tile that leaves the row, frames-per-workgroup granularity
    L = fsea.hip_lib()
    bad = [(nf - 1, shape[0], shape[1], first_x, tile_rows, step), (nf, shape[0], shape[1], first_x, tile_rows, n - 4),
           (nf, shape[0], shape[1], first_x + 64, tile_rows, step), (nf, tile_rows - 1, shape[1], first_x, tile_rows, step),
           (nf, shape[0], shape[1], first_x + 1, tile_rows, step)]
    for frames, ir, st, fx, tr, ts in bad:
        assert L.fsea_exec_u8_tiled_device(plan._p, d_in.ptr, frames, 1, d_img.ptr, ir, st, fx, tr, ts, None) != 0
    if n == 4096:
        assert L.fsea_exec_u8_tiled_device(plan._p, d_in.ptr, 5, 1, d_img.ptr, shape[0], shape[1], first_x, 1, step, None) != 0
    for b in (d_in, d_rows, d_img):
        b.free()
    plan.close()


def test_device_resident_and_ragged_counts():
    n = 2048
    plan = fsea.Plan(n)
    for nf in (1, 3, 4, 5, 1023):
        iq = synth_iq(nf, 2 * nf * n)
        d_in = DeviceBuffer(iq.nbytes).upload(iq)
        d_out = DeviceBuffer(nf * n * 4)
        plan.exec_device(d_in.ptr, nf, d_out.ptr)
        plan.synchronize()
        got = d_out.download(np.float32, (nf, n))
        parity.check_mode(got, iq, n, nf, n, True, 0)
        d_in.free()
        d_out.free()
    plan.close()


def test_repeated_launches_reset_their_ticket_counters(units_policy):
    """Frames are handed out by atomic ticket counters that the last workgroup of a launch resets;
    a stale counter would make a later launch skip frames.  Launch the same plan many times with
    changing frame counts (more than the 64 counter slots) and check every row each time."""
    n = 1024
    plan = fsea.Plan(n)
    nf_max = 5000
    iq = synth_iq(77, 2 * nf_max * n)
    want = O.rows(iq, 64, n)
    d_in = DeviceBuffer(iq.nbytes).upload(iq)
    d_out = DeviceBuffer(nf_max * n * 4)
    rng = np.random.default_rng(0)
    for it in range(150):
        nf = int(rng.integers(1, nf_max))
        d_out.upload(np.full(min(nf, 64) * n, -1.0, np.float32))      # poison what will be checked
        plan.exec_device(d_in.ptr, nf, d_out.ptr)
        plan.synchronize()
        k = min(nf, 64)
        got = d_out.download(np.float32, (k, n))
        assert got.min() >= 0.0, "launch %d (nf=%d) left rows unwritten" % (it, nf)
        parity.check_float(got, want[:k])
    # last rows of a big launch too
    plan.exec_device(d_in.ptr, nf_max, d_out.ptr)
    plan.synchronize()
    got = d_out.download(np.float32, (nf_max, n))[-8:]
    parity.check_float(got, O.rows(iq[2 * n * (nf_max - 8):], 8, n))
    d_in.free()
    d_out.free()
    plan.close()


def test_zero_frames_and_single_frame():
    plan = fsea.Plan(4096)
    assert plan.exec_host(np.zeros(0, np.uint8), 0).shape == (0, 4096)
    d = DeviceBuffer(4096 * 4)
    plan.exec_device(d.ptr, 0, d.ptr)                    # no launch, no error
    iq = synth_iq(1, 2 * 4096)
    parity.check_mode(plan.exec_host(iq, 1), iq, 4096, 1, 4096, True, 0)
    d.free()
    plan.close()


def test_concurrent_launches_of_one_plan_on_two_streams(units_policy):
    """Every launch draws its own ticket-counter slot, so launches of one plan may overlap on
    different streams.  Two HIP streams, interleaved launches into separate outputs."""
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipStreamCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    streams = [ctypes.c_void_p(), ctypes.c_void_p()]
    for st in streams:
        assert hip.hipStreamCreate(ctypes.byref(st)) == 0
    n, nf = 8192, 700
    plan = fsea.Plan(n)
    iqs = [synth_iq(900 + k, 2 * nf * n) for k in range(2)]
    d_in = [DeviceBuffer(iq.nbytes).upload(iq) for iq in iqs]
    d_out = [DeviceBuffer(nf * n * 4) for _ in range(2)]
    for rep in range(40):
        for k in range(2):
            plan.exec_device(d_in[k].ptr, nf, d_out[k].ptr, stream=streams[k].value)
    for st in streams:
        assert hip.hipStreamSynchronize(st) == 0
    for k in range(2):
        got = d_out[k].download(np.float32, (nf, n))
        for f in (0, 1, nf // 2, nf - 2, nf - 1):
            parity.check_float(got[f], O.rows(iqs[k][2 * f * n: 2 * (f + 1) * n], 1, n)[0])
        d_in[k].free()
        d_out[k].free()
    for st in streams:
        hip.hipStreamDestroy(st)
    plan.close()


def test_many_overlapping_launches_on_four_streams_and_reset(units_policy):
    """One ticket-counter slot per stream (launches on a stream run in order; streams never share a
    slot): far more than 64 launches in flight over four streams, multi-wave size, every launch
    checked.  Then fsea_plan_reset, and the plan still works on a fifth stream."""
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    streams = [ctypes.c_void_p() for _ in range(5)]
    for st in streams:
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0          # hipStreamNonBlocking
    n, nf, reps = 4096, 300, 48                                               # 4 x 48 = 192 launches queued
    plan = fsea.Plan(n)
    iq = synth_iq(123, 2 * nf * n)
    want = O.rows(iq, nf, n)
    d_in = DeviceBuffer(iq.nbytes).upload(iq)
    outs = [[DeviceBuffer(nf * n * 4) for _ in range(reps)] for _ in range(4)]
    for r in range(reps):
        for k in range(4):
            plan.exec_device(d_in.ptr, nf - (r % 3), outs[k][r].ptr, stream=streams[k].value)
    for st in streams[:4]:
        assert hip.hipStreamSynchronize(st) == 0
    for k in range(4):
        for r in range(reps):
            rows = nf - (r % 3)
            got = outs[k][r].download(np.float32, (rows, n))
            parity.check_float(got, want[:rows])
            outs[k][r].free()
    plan.reset()
    d_out = DeviceBuffer(nf * n * 4)
    plan.exec_device(d_in.ptr, nf, d_out.ptr, stream=streams[4].value)
    assert hip.hipStreamSynchronize(streams[4]) == 0
    parity.check_float(d_out.download(np.float32, (nf, n)), want)
    d_in.free()
    d_out.free()
    for st in streams:
        hip.hipStreamDestroy(st)
    plan.close()


def test_a_plan_outlives_any_number_of_short_lived_streams():
    """ticket-counter slots are recycled once the launch they served has completed, and launches that do not
    use the counters (single-wave sizes, short launches) take none: a plan launched on far more than 64 distinct,
    short-lived streams over its lifetime keeps working, with no device-wide synchronisation in between."""
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    for n, nf, policy in ((8192, 600, fsea.UNITS_TICKETS), (4096, 2100, fsea.UNITS_TICKETS), (1024, 300, fsea.UNITS_AUTO)):
        plan = fsea.Plan(n)
        plan.set_unit_distribution(policy)
        iq = synth_iq(31 + n, 2 * nf * n)
        want = O.rows(iq, 4, n)
        want_last = O.rows(iq[2 * n * (nf - 2):], 2, n)
        d_in = DeviceBuffer(iq.nbytes).upload(iq)
        d_out = DeviceBuffer(nf * n * 4)
        live = []
        for k in range(150):
            st = ctypes.c_void_p()
            assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
            plan.exec_device(d_in.ptr, nf, d_out.ptr, stream=st.value)
            live.append(st)
            if len(live) == 3:                     # a few streams alive at a time; handles get recycled by the runtime
                old = live.pop(0)
                assert hip.hipStreamSynchronize(old) == 0
                hip.hipStreamDestroy(old)
            if k % 37 == 0:
                assert hip.hipStreamSynchronize(st) == 0
                got = d_out.download(np.float32, (nf, n))
                parity.check_float(got[:4], want)
                parity.check_float(got[-2:], want_last)
        for st in live:
            assert hip.hipStreamSynchronize(st) == 0
            hip.hipStreamDestroy(st)
        got = d_out.download(np.float32, (nf, n))
        parity.check_float(got[:4], want)
        parity.check_float(got[-2:], want_last)
        d_in.free()
        d_out.free()
        plan.close()


@pytest.mark.parametrize("n,nf", [(16384, 2000), (16384, 7), (8192, 4097), (8192, 2), (8192, 300)])
def test_half_overlap_kernel_equals_the_ordinary_kernel(n, nf, monkeypatch):
    """hop == N/2 with MAG rows of raw int8 input runs `*_u8_mag_half` at 8192 and 16384 points: runs of consecutive frames
    per workgroup, the second half of each frame's bytes kept in registers for the next frame.  Same rows, bit for bit,
    as the ordinary kernel (FSEA_NO_HALF_OVERLAP=1 at plan creation), and both against the oracle."""
    hop = n // 2
    iq = synth_iq(n + nf, 2 * ((nf - 1) * hop + n))
    d_in = DeviceBuffer(iq.nbytes).upload(iq)
    rows = {}
    for half in (True, False):
        if half:
            monkeypatch.delenv("FSEA_NO_HALF_OVERLAP", raising=False)
        else:
            monkeypatch.setenv("FSEA_NO_HALF_OVERLAP", "1")
        plan = fsea.Plan(n, hop=hop)
        assert plan.kernel_name == ("fsea_fft%d_u8_mag%s" % (n, "_half" if half else ""))
        d_out = DeviceBuffer(nf * n * 4)
        d_out.upload(np.full(nf * n, -1.0, np.float32))
        plan.exec_device(d_in.ptr, nf, d_out.ptr)
        plan.exec_device(d_in.ptr, nf, d_out.ptr)
        plan.synchronize()
        rows[half] = d_out.download(np.float32, (nf, n))
        d_out.free()
        plan.close()
    assert np.array_equal(rows[True], rows[False])
    for f in sorted({0, 1, nf // 2, nf - 1}):
        parity.check_mode(rows[True][f:f + 1], iq[2 * f * hop:], n, 1, hop, True, 0)
    d_in.free()


def test_per_thread_default_stream_from_two_threads():
    """hipStreamPerThread is ONE handle that names a different stream in every host thread.  Launches on it never
    keep a ticket-counter slot (each takes a free one, recycled by event), so two threads launching long launches of one
    plan on their per-thread streams at the same time do not share a slot."""
    import threading
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    per_thread = 2                                   # hipStreamPerThread
    n, nf = 8192, 1200
    plan = fsea.Plan(n)
    plan.set_unit_distribution(fsea.UNITS_TICKETS)
    iqs = [synth_iq(60 + k, 2 * nf * n) for k in range(2)]
    d_in = [DeviceBuffer(iq.nbytes).upload(iq) for iq in iqs]
    d_out = [DeviceBuffer(nf * n * 4) for _ in range(2)]
    errors = []

    def work(k):
        try:
            for _ in range(25):
                plan.exec_device(d_in[k].ptr, nf, d_out[k].ptr, stream=per_thread)
            assert hip.hipStreamSynchronize(ctypes.c_void_p(per_thread)) == 0
        except Exception as e:                      # pragma: no cover - reported below
            errors.append(e)
    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(2):
        got = d_out[k].download(np.float32, (nf, n))
        for f in (0, 1, nf // 2, nf - 1):
            parity.check_float(got[f], O.rows(iqs[k][2 * f * n: 2 * (f + 1) * n], 1, n)[0])
        d_in[k].free()
        d_out[k].free()
    plan.close()


def test_host_path_from_two_threads_sharing_one_input_buffer():
    """The pipelined host path pins the caller's pages in place for the call; two host threads handing the SAME capture to
    two plans at once share one counted registration (the first to finish must not unpin under the other)."""
    import threading
    n_a, n_b = 1024, 2048
    frames = 6000                                   # 12-24 MiB in, 24-48 MiB out: the chunked, three-stream path
    iq = synth_iq(808, 2 * frames * n_b)
    plans = {n_a: fsea.Plan(n_a), n_b: fsea.Plan(n_b)}
    outs = {n: [] for n in plans}
    errors = []

    def work(n):
        try:
            for _ in range(6):
                outs[n].append(plans[n].exec_host(iq, frames))
        except Exception as e:                      # pragma: no cover - reported below
            errors.append(e)
    threads = [threading.Thread(target=work, args=(n,)) for n in plans]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for n, got in outs.items():
        for g in got[1:]:
            assert np.array_equal(g, got[0])
        for f in (0, frames // 2, frames - 1):
            parity.check_mode(got[0][f:f + 1], iq[2 * f * n:], n, 1, n, True, 0)
        plans[n].close()


def test_calls_leave_the_current_device_alone():
    """Every entry point runs on the plan's device and restores the caller's current HIP device
    (ADVICE r01); with one GPU the observable part is that it stays 0 and nothing fails."""
    hip = ctypes.CDLL("libamdhip64.so")
    dev = ctypes.c_int(-1)
    assert hip.hipGetDevice(ctypes.byref(dev)) == 0
    before = dev.value
    plan = fsea.Plan(256)
    iq = synth_iq(9, 2 * 4 * 256)
    plan.exec_host(iq, 4)
    assert hip.hipGetDevice(ctypes.byref(dev)) == 0 and dev.value == before
    plan.close()


@pytest.mark.parametrize("n,nf", [(1024, 7), (4096, 40)])      # mapped-staging path / copy path (> 256 KiB)
def test_f64_input_branch(n, nf):
    rng = np.random.default_rng(3)
    x = rng.normal(0, 0.2, 2 * nf * n)
    plan = fsea.Plan(n)
    got = plan.exec_host_f64(x, nf)
    want = np.stack([O.mag_row(O.fft_forward(O.unpack_center_f64(x[2 * f * n: 2 * (f + 1) * n])))
                     for f in range(nf)])
    parity.check_float(got, want)
    plan.close()


def test_invalid_arguments_are_reported():
    plan = fsea.Plan(1024)
    L = fsea.hip_lib()
    d = DeviceBuffer(4096)
    assert L.fsea_exec_u8_device(plan._p, d.ptr.value + 2, 1, 1, d.ptr, None) == -1
    assert b"aligned" in L.fsea_last_error_string()
    assert L.fsea_exec_u8_device(plan._p, None, 1, 1, d.ptr, None) == -1
    d.free()
    plan.close()


@pytest.mark.parametrize("n,nf", [(8192, 4096), (1024, 32768), (4096, 8192)])
def test_full_size_properties(n, nf):
    """BASELINE.json sizes: Parseval on every row, tone position, and EVERY row against the oracle (its frames sharded
    over the host's cores, oracle.rows_mt: the single-threaded rows bit for bit)."""
    iq = synth_iq(3, 2 * nf * n)                      # seed 3 = SURVEY 8(d) config C3
    d_in = DeviceBuffer(iq.nbytes).upload(iq)
    d_out = DeviceBuffer(nf * n * 4)
    plan = fsea.Plan(n, mode=fsea.MODE_MAG_NODC_F32)
    plan.exec_device(d_in.ptr, nf, d_out.ptr)
    plan.synchronize()
    mag = d_out.download(np.float32, (nf, n)).astype(np.float64)
    u = (iq ^ np.uint8(0x80)).astype(np.float64).reshape(nf, 2 * n) / 256.0
    energy_in = n * np.sum(u * u, axis=1)
    energy_out = np.sum(mag * mag, axis=1)
    assert np.max(np.abs(energy_out - energy_in) / energy_in) < 2e-6          # Parseval per row
    # the +fs/8 tone lands at bin N/2 + N/8 in every row (below only the DC bin N/2)
    no_dc = mag.copy()
    no_dc[:, n // 2] = 0
    assert np.all(no_dc.argmax(axis=1) == n // 2 + n // 8)
    plan.close()
    plan = fsea.Plan(n)
    plan.exec_device(d_in.ptr, nf, d_out.ptr)
    plan.synchronize()
    got = d_out.download(np.float32, (nf, n))
    want = O.rows_mt(iq, nf, n)
    slab = max(1, (1 << 22) // n)                       # the tolerance is per row set: slabs keep the L2 figure local
    for f0 in range(0, nf, slab):
        parity.check_float(got[f0:f0 + slab], want[f0:f0 + slab])
    del want
    # MAG == MAG_NODC everywhere except the patched bin: bit for bit where the two modes run the same configuration of the
    # size, within the rounding of two radix orders where they do not (1024 points: frequensea_amd/csrc/fsea_configs.h)
    keep = np.ones(n, bool)
    keep[n // 2] = False
    if kernel_stem(n, fsea.MODE_MAG_F32) == kernel_stem(n, fsea.MODE_MAG_NODC_F32):
        assert np.array_equal(got[:, keep], mag[:, keep].astype(np.float32))
    else:
        assert np.max(np.abs(got[:, keep] - mag[:, keep])) <= 2e-6 * mag.max()
    d_in.free()
    d_out.free()
    plan.close()


def test_mean_magnitude_gate_and_composite():
    n, nf = 256, 100                                    # c/fft-batch-broad.c: 256-pt, 100-row gate
    iq = synth_iq(9, 2 * nf * n)
    d_in = DeviceBuffer(iq.nbytes).upload(iq)
    plan = fsea.Plan(n, mode=fsea.MODE_DB5_U8_DCFIX)
    mean = plan.mean_magnitude_device(d_in.ptr, nf)
    want = O.mean_magnitude(O.rows(iq, nf, n, mode=O.MODE_COMPLEX))
    assert abs(mean - want) <= 2e-6 * want
    # tiles -> stitched image, 50 % overlap as c/fft-stitch.c (bit-exact integer max)
    tiles = [np.random.default_rng(k).integers(0, 256, (64, n), dtype=np.uint8) for k in range(3)]
    width = n + 2 * (n // 2)
    want_img = np.zeros((64, width), np.uint8)
    d_img = DeviceBuffer(want_img.nbytes).upload(want_img)
    for k, t in enumerate(tiles):
        O.composite_max(want_img, t, k * (n // 2))
        d_t = DeviceBuffer(t.nbytes).upload(t)
        fsea.composite_max_device(d_img.ptr, d_t.ptr, k * (n // 2), 0, n, 64, width, 64, n)
        with pytest.raises(fsea.FseaError, match="do not fit"):          # a tile reaching past the last row
            fsea.composite_max_device(d_img.ptr, d_t.ptr, 0, 1, n, 64, width, 64, n)
        with pytest.raises(fsea.FseaError, match="row stride"):          # or past the row, without 32-bit wrap-around
            fsea.composite_max_device(d_img.ptr, d_t.ptr, 0xFFFFFF00, 0, n, 64, width, 64, n)
        d_t.free()                                      # hipFree synchronises the device
    got_img = d_img.download(np.uint8, want_img.shape)
    assert np.array_equal(got_img, want_img)
    # the same stitch as one call over the contiguous tile stack (50 % overlap -> two phases)
    d_img2 = DeviceBuffer(want_img.nbytes).upload(np.zeros_like(want_img))
    d_stack = DeviceBuffer(3 * 64 * n).upload(np.stack(tiles))
    fsea.stitch_tiles_device(d_img2.ptr, d_stack.ptr, 3, 0, n // 2, n, 64, width)
    assert np.array_equal(d_img2.download(np.uint8, want_img.shape), want_img)
    d_img2.free()
    d_stack.free()
    d_in.free()
    d_img.free()
    plan.close()


# ---- the reference-shaped API (what lua/fft.lua drives through src/main.cpp) -------------

def _nut_u8(L, arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    return L.nut_buffer_new_u8(arr.size // 2, 2, arr.ctypes.data)


@pytest.fixture(params=["host", "device"])
def history_mode(request, monkeypatch):
    """nrf_fft's history: the host ring (default) or the device-resident ring (NRF_FFT_HISTORY=device,
    fsea_history_*); nrf_fft_new reads the variable.  Both must give identical buffers."""
    monkeypatch.setenv("NRF_FFT_HISTORY", request.param)
    return request.param


def test_nrf_fft_api_history_and_shift(golden, history_mode):
    L = nrf.nrf_lib()
    n, h = 256, 4
    fft = L.nrf_fft_new(n, h)
    keys = GOLDEN_KEYS
    for key in keys:                                    # four process calls, newest row first
        flipped = golden[key + "__flipped"]             # device buffers are offset binary
        buf = _nut_u8(L, flipped)
        L.nrf_fft_process(fft, buf)
        L.nut_buffer_free(buf)
    out = L.nrf_fft_get_buffer(fft)
    c = out.contents
    assert (c.type, c.length, c.channels, c.size_bytes) == (nrf.NUT_BUFFER_F64, n * h, 1, 8 * n * h)
    hist = nrf.buffer_to_numpy(L, out).reshape(h, n)
    L.nut_buffer_free(out)
    want = np.stack([golden[k + "__mag_256"] for k in reversed(keys)])
    parity.check_float(hist, want)
    # shifts: compare with the oracle applied to the same history (exact copies of doubles)
    for d in (8.0, -8.0, 50.0, -3.0, 0.5, 1e9):
        ref = hist.copy() if d == 8.0 else ref
        O.fft_shift(ref, n, h, d)
        L.nrf_fft_shift(fft, d)
        out = L.nrf_fft_get_buffer(fft)
        now = nrf.buffer_to_numpy(L, out).reshape(h, n)
        L.nut_buffer_free(out)
        assert np.array_equal(now, ref), d
    L.nrf_fft_free(fft)


def test_nrf_fft_lua_sizes_and_f64_input(golden, history_mode):
    """The sizes the shipped Lua scenes use: (1024,1024) fft.lua:34, (128,512) fft-sea.lua:211."""
    L = nrf.nrf_lib()
    raw = golden["rf_202p500_2__flipped"]
    for n, h in ((1024, 1024), (128, 512)):
        fft = L.nrf_fft_new(n, h)
        buf = _nut_u8(L, raw)
        for _ in range(3):
            L.nrf_fft_process(fft, buf)
        out = L.nrf_fft_get_buffer(fft)
        hist = nrf.buffer_to_numpy(L, out).reshape(h, n)
        L.nut_buffer_free(out)
        want = golden["rf_202p500_2__mag_%d" % n]
        for r in range(3):
            parity.check_float(hist[r], want)
        assert not hist[3:].any()
        # f64 input (nrf_freq_shifter output feeds nrf_fft in lua/fft-shifted.lua:52-55)
        f64 = L.nut_buffer_convert(buf, nrf.NUT_BUFFER_F64)
        L.nrf_fft_process(fft, f64)
        out = L.nrf_fft_get_buffer(fft)
        parity.check_float(nrf.buffer_to_numpy(L, out).reshape(h, n)[0], want)
        L.nut_buffer_free(out)
        L.nut_buffer_free(f64)
        L.nut_buffer_free(buf)
        L.nrf_fft_free(fft)


def _expected_rows_any_size(iq, n, nf, hop, flip, mode, exact):
    """Rows for a transform size the oracle's power-of-two FFT does not take: the oracle's own loops around its O(n^2)
    long-double DFT (`exact`, small n) or around numpy's f64 FFT (large n: an O(n^2) check there would take minutes)."""
    rows = []
    for f in range(nf):
        raw = iq[2 * f * hop: 2 * (f * hop + n)]
        x = O.unpack_center_u8(O.flip_u8(raw) if flip else raw)
        spec = O.dft_naive(x) if exact else np.fft.fft(x)
        rows.append({0: lambda: O.mag_row(spec), 1: lambda: O.db_u8_row(spec, 10.0, 0), 2: lambda: O.db_u8_row(spec, 5.0, 1),
                     3: lambda: spec, 4: lambda: np.abs(spec)}[mode]())
    return np.stack(rows)


@pytest.mark.parametrize("n", [1000, 1001, 96, 17, 16, 2, 3000, 8191])
def test_transform_sizes_fftw_takes_and_the_kernels_do_not(n):
    """`fftw_plan_dft_1d` (src/nrf.c:564) takes any size; the gfx950 kernels are powers of two from 32 to 16384.  Every
    other size from 2 to 8192 runs through Bluestein's algorithm on those kernels (two transforms of size 2^p >= 2n - 1
    around a pointwise product with the chirp's spectrum; the offset-binary DC term restored from a table, for odd n in
    every bin).  Same tolerance as the power-of-two sizes; bin n/2 := bin n/2 - 1 with the integer n/2 of the reference."""
    nf, exact = 3, n <= 1001
    for hop in (n, max(1, n // 3)):
        iq = synth_iq(n + hop, 2 * ((nf - 1) * hop + n))
        for mode in (0, 3, 1, 2, 4):
            for flip in ((True, False) if mode in (0, 1) else (True,)):
                plan = fsea.Plan(n, hop=hop, mode=mode)
                assert plan.kernel_name.startswith("bluestein(fsea_fft")
                got = plan.exec_host(iq, nf, flip=flip)
                plan.close()
                want = _expected_rows_any_size(iq, n, nf, hop, flip, mode, exact)
                if mode in (1, 2):
                    parity.check_u8(got, want)
                else:
                    parity.check_float(got, want)
                if mode in (0, 2) and n >= 4:
                    assert np.array_equal(got[:, n // 2], got[:, n // 2 - 1])
    # resident data, a batch larger than the work buffers' chunk, f64 input, and the entry points that do not exist
    nf = 70000 if n == 17 else 9
    iq = synth_iq(5 * n, 2 * nf * n)
    plan = fsea.Plan(n)
    d_in = DeviceBuffer(iq.nbytes).upload(iq)
    d_out = DeviceBuffer(nf * n * 4)
    plan.exec_device(d_in.ptr, nf, d_out.ptr)
    plan.synchronize()
    got = d_out.download(np.float32, (nf, n))
    for f in (0, nf // 2, nf - 1):
        parity.check_float(got[f:f + 1], _expected_rows_any_size(iq[2 * f * n:], n, 1, n, True, 0, exact))
    x = np.random.default_rng(n).normal(0, 0.3, 2 * 2 * n)
    got = plan.exec_host_f64(x, 2)
    for f in range(2):
        spec = np.fft.fft(O.unpack_center_f64(x[2 * f * n: 2 * (f + 1) * n]))
        parity.check_float(got[f], O.mag_row(spec))
    with pytest.raises(fsea.FseaError, match="Bluestein"):
        plan.exec_shifted_device(d_in.ptr, 1, d_out.ptr, 0.01)
    d_in.free()
    d_out.free()
    plan.close()


@pytest.mark.parametrize("n", [32768, 65536, 1 << 18, 9000, 20000])
def test_transform_sizes_above_the_largest_kernel(n):
    """Powers of two above 16384 run as two passes of the kernels (four-step: n = n1 n2, column transforms, twiddle,
    row transforms; up to 2^20 points), and sizes above 8192 that are not powers of two through Bluestein's algorithm on
    top of that.  Against numpy's f64 FFT around the oracle's unpack / magnitude / pixel loops (an O(n^2) DFT of this
    size takes minutes); same tolerance as every other size."""
    nf = 2
    iq = synth_iq(n, 2 * nf * n)
    for mode in (0, 3, 2):
        plan = fsea.Plan(n, mode=mode)
        assert plan.kernel_name.startswith(("fourstep(", "bluestein(fourstep("))
        got = plan.exec_host(iq, nf)
        plan.close()
        want = _expected_rows_any_size(iq, n, nf, n, True, mode, False)
        if mode == 2:
            parity.check_u8(got, want)
            assert np.array_equal(got[:, n // 2], got[:, n // 2 - 1])
        else:
            parity.check_float(got, want)
    if n == 32768:                       # offset-binary bytes, f64 input, resident data
        plan = fsea.Plan(n, mode=3)
        got = plan.exec_host(iq ^ np.uint8(0x80), nf, flip=False)
        parity.check_float(got, _expected_rows_any_size(iq, n, nf, n, True, 3, False))
        x = np.random.default_rng(1).normal(0, 0.3, 2 * n)
        parity.check_float(plan.exec_host_f64(x, 1)[0], np.fft.fft(O.unpack_center_f64(x)))
        plan.close()


def test_nrf_fft_with_a_size_that_is_not_a_power_of_two(history_mode):
    """nrf_fft_new(1000, 4) as a Lua script could ask for (FFTW plans any size): process / get_buffer / shift through the
    reference's API, both history modes."""
    L = nrf.nrf_lib()
    n, h = 1000, 4
    fft = L.nrf_fft_new(n, h)
    raws = [synth_iq(40 + k, nrf.NRF_BUFFER_SIZE_BYTES) ^ np.uint8(0x80) for k in range(3)]   # device buffers are offset binary
    for raw in raws:
        buf = _nut_u8(L, raw)
        L.nrf_fft_process(fft, buf)
        L.nut_buffer_free(buf)
    out = L.nrf_fft_get_buffer(fft)
    hist = nrf.buffer_to_numpy(L, out).reshape(h, n)
    L.nut_buffer_free(out)
    want = np.stack([O.mag_row(O.dft_naive(O.unpack_center_u8(raw[: 2 * n]))) for raw in reversed(raws)])
    parity.check_float(hist[:3], want)
    assert not hist[3].any()
    ref = hist.copy()
    O.fft_shift(ref, n, h, 7.0)
    L.nrf_fft_shift(fft, 7.0)
    out = L.nrf_fft_get_buffer(fft)
    assert np.array_equal(nrf.buffer_to_numpy(L, out).reshape(h, n), ref)
    L.nut_buffer_free(out)
    L.nrf_fft_free(fft)


def test_block_graph_device_to_fft(tmp_path):
    """nrf_block_connect(device, fft): the replay thread pushes rows (src/nrf.c:37-50,118)."""
    import time
    L = nrf.nrf_lib()
    raw = synth_iq(21, nrf.NRF_BUFFER_SIZE_BYTES)
    path = tmp_path / "one.raw"
    raw.tofile(path)
    fft = L.nrf_fft_new(1024, 8)
    dev = L.nrf_device_new(100.0, str(path).encode())
    L.nrf_block_connect(dev, fft)
    time.sleep(0.3)
    L.nrf_device_free(dev)
    out = L.nrf_fft_get_buffer(fft)
    hist = nrf.buffer_to_numpy(L, out).reshape(8, 1024)
    L.nut_buffer_free(out)
    want = O.rows(raw, 1, 1024)[0]
    assert hist[0].any()
    parity.check_float(hist[0], want)
    L.nrf_fft_free(fft)


# ---------------------------------------------------------------------------------------------
# frequency shifter in front of the FFT (src/nrf.c:843-866; lua/fft-shifted.lua:52-55)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", SIZES)
def test_frequency_shifted_rows_all_sizes(n):
    """fsea_exec_u8_shifted_host: the shifter fused into the FFT kernel's load vs the oracle's
    shifter -> F64-branch FFT.  150 kHz at 5 Msps, the step lua/fft-shifted.lua works in."""
    nf = 150 if n <= 1024 else 21
    iq = synth_iq(600 + n, 2 * nf * n)
    delta, phase0 = 150000 / 5000000, 0.8125
    for mode in (0, 3, 1, 5):
        plan = fsea.Plan(n, mode=mode)
        got = plan.exec_shifted_host(iq, nf, delta, phase0)
        parity.check_mode_shifted(got, iq, n, nf, n, True, mode, delta, phase0)
        plan.close()


@pytest.mark.parametrize("n,hop,delta,flip", [(1024, 256, -0.123456789, True), (4096, 2048, 0.49, False),
                                              (8192, 8192, 0.0, True), (16384, 8192, 7.25, True),
                                              (256, 8, 1e-7, True)])
def test_frequency_shift_geometry_and_sign(n, hop, delta, flip):
    nf = 11
    iq = synth_iq(700 + n, 2 * ((nf - 1) * hop + n))
    plan = fsea.Plan(n, hop=hop, mode=fsea.MODE_COMPLEX_F32)
    got = plan.exec_shifted_host(iq, nf, delta, 0.0, flip=flip)
    parity.check_mode_shifted(got, iq, n, nf, hop, flip, 3, delta, 0.0)
    plan.close()


def test_frequency_shift_by_whole_bins_rolls_the_spectrum_at_full_size():
    """Size-independent property at the BASELINE batch (8192 points, 4096 frames): shifting by
    k bins rotates the unshifted complex spectrum by k and adds the shifter's 0.5 (1+i) to bin N/2;
    a stream continued with phase0 equals the same stream processed in one call."""
    n, nf, k = 8192, 4096, 1237
    iq = synth_iq(3, 2 * nf * n)
    d_in = DeviceBuffer(iq.nbytes).upload(iq)
    d_a, d_b = DeviceBuffer(nf * n * 8), DeviceBuffer(nf * n * 8)
    plan = fsea.Plan(n, mode=fsea.MODE_COMPLEX_F32)
    plan.exec_device(d_in.ptr, nf, d_a.ptr)
    plan.exec_shifted_device(d_in.ptr, nf, d_b.ptr, k / n)
    plan.synchronize()
    rows = np.unique(np.r_[0, 1, nf - 1, np.random.default_rng(9).integers(0, nf, 40)])
    plain = d_a.download(np.complex64, (nf, n))[rows].astype(np.complex128)
    shifted = d_b.download(np.complex64, (nf, n))[rows].astype(np.complex128)
    want = np.roll(plain, k, axis=1)
    want[:, n // 2] += 0.5 * n * (1 + 1j) - 0.0
    # the unshifted path restores its own offset-binary DC in bin n/2; after the roll that sits in
    # bin n/2 + k, exactly where the shifter moves it
    err = np.linalg.norm(shifted - want, axis=1) / np.linalg.norm(want, axis=1)
    assert err.max() < 1e-6, err.max()
    # second half of the batch as a continued stream: phase0 = samples consumed * cycles per sample
    half = nf // 2
    delta = 150000 / 5000000
    plan.exec_shifted_device(d_in.ptr, nf, d_a.ptr, delta)
    plan.exec_shifted_device(d_in.ptr.value + 2 * half * n, nf - half, d_b.ptr, delta, (half * n) * delta)
    plan.synchronize()
    whole = d_a.download(np.complex64, (nf, n))[half:]
    cont = d_b.download(np.complex64, (nf, n))[: nf - half]
    err = np.linalg.norm(whole - cont, axis=1) / np.linalg.norm(whole, axis=1)
    assert err.max() < 1e-6, err.max()
    for f in (0, nf - half - 1):
        want = O.rows_shifted(iq[2 * (half + f) * n: 2 * (half + f + 1) * n], 1, n, delta, ((half + f) * n) * delta,
                              mode=O.MODE_COMPLEX)[0]
        parity.check_float(cont[f], want)
    for b in (d_in, d_a, d_b):
        b.free()
    plan.close()


def test_lua_fft_shifted_call_chain(tmp_path):
    """lua/fft-shifted.lua:52-55 through the reference-shaped C API: device block -> nrf_freq_shifter
    (host, f64) -> nrf_fft (GPU, F64 input branch); and the same rows from the fused device path."""
    L = nrf.nrf_lib()
    raw = synth_iq(41, 2 * nrf.NRF_BUFFER_SIZE_BYTES)
    blocks = (raw ^ np.uint8(0x80)).reshape(2, -1)
    n, h, off, fs = 1024, 16, 150000, 5000000
    fft = L.nrf_fft_new(n, h)
    sh = L.nrf_freq_shifter_new(off, fs)
    for blk in blocks:
        buf = L.nut_buffer_new_u8(nrf.NRF_SAMPLES_LENGTH, 2, blk.ctypes.data)
        L.nrf_freq_shifter_process(sh, buf)
        shifted = L.nrf_freq_shifter_get_buffer(sh)
        L.nrf_fft_process(fft, shifted)
        L.nut_buffer_free(shifted)
        L.nut_buffer_free(buf)
    out = L.nrf_fft_get_buffer(fft)
    hist = nrf.buffer_to_numpy(L, out).reshape(h, n)
    L.nut_buffer_free(out)
    L.nrf_freq_shifter_free(sh)
    L.nrf_fft_free(fft)
    plan = fsea.Plan(n)
    for b, blk in enumerate(blocks):
        phase0 = b * nrf.NRF_SAMPLES_LENGTH * (off / fs)          # what the shifter's state amounts to
        want = O.rows_shifted(blk[: 2 * n], 1, n, off / fs, phase0, flip=False)[0]
        parity.check_float(hist[1 - b], want)                     # newest row first
        fused = plan.exec_shifted_host(blk[: 2 * n], 1, off / fs, phase0, flip=False)[0]
        parity.check_float(fused, want)
    plan.close()


@pytest.mark.parametrize("n", [1024, 8192])
def test_batches_beyond_4_gib(n):
    """Maximum sizes: one launch whose input (4.5 GiB) and output (9 GiB) both cross 32-bit byte
    offsets (static frame interleave at 1024, ticket pools at 8192).  The kernel rebases a 64-bit
    buffer window per unit; rows on either side of every 4 GiB boundary, the first and the last are
    compared with the oracle."""
    chunk_frames = (1 << 25) // n                            # 64 MiB of IQ per host chunk
    n_chunks = 72
    nf = chunk_frames * n_chunks                             # 4.5 GiB in, 9 GiB out
    L = fsea.hip_lib()
    chunk = synth_iq(4242, 2 * chunk_frames * n)
    d_in = DeviceBuffer(chunk.nbytes * n_chunks)
    d_out = DeviceBuffer(nf * n * 4)
    for k in range(n_chunks):                                # the same 64 MiB repeated; frame f = chunk frame f % 32768
        fsea._check(L.fsea_copy_to_device(0, ctypes.c_void_p(d_in.ptr.value + k * chunk.nbytes),
                                          chunk.ctypes.data, chunk.nbytes))
    plan = fsea.Plan(n)
    plan.exec_device(d_in.ptr, nf, d_out.ptr)
    plan.synchronize()
    in_edge = (1 << 32) // (2 * n)                           # first frame past 4 GiB of input
    out_edges = [(1 << 32) // (4 * n), (2 << 32) // (4 * n)]  # first frames past 4 and 8 GiB of output
    frames = sorted({0, 1, nf - 1, nf - 2, in_edge - 1, in_edge, in_edge + 1,
                     *[e + d for e in out_edges for d in (-1, 0, 1)]})
    row = np.empty(n, np.float32)
    for f in frames:
        fsea._check(L.fsea_copy_to_host(0, row.ctypes.data, ctypes.c_void_p(d_out.ptr.value + f * n * 4), n * 4))
        g = f % chunk_frames
        parity.check_float(row, O.rows(chunk[2 * g * n: 2 * (g + 1) * n], 1, n)[0])
    # and a sample of rows against their periodic copies (every chunk holds the same frames)
    a = np.empty((64, n), np.float32)
    b = np.empty((64, n), np.float32)
    fsea._check(L.fsea_copy_to_host(0, a.ctypes.data, ctypes.c_void_p(d_out.ptr.value + 100 * n * 4), a.nbytes))
    fsea._check(L.fsea_copy_to_host(0, b.ctypes.data,
                                    ctypes.c_void_p(d_out.ptr.value + (71 * chunk_frames + 100) * n * 4), b.nbytes))
    assert np.array_equal(a, b)
    d_in.free()
    d_out.free()
    plan.close()


def test_pixel_modes_at_full_size():
    """BASELINE batch (8192 points x 4096 frames) through the u8 dB epilogues: every row against the
    oracle (exact on >= 99.9 % of pixels, +-1 elsewhere, per slab of 64 rows) and the DC-fix column of the broad mode."""
    n, nf = 8192, 4096
    iq = synth_iq(3, 2 * nf * n)
    d_in = DeviceBuffer(iq.nbytes).upload(iq)
    d_out = DeviceBuffer(nf * n)
    for mode in (fsea.MODE_DB10_U8, fsea.MODE_DB5_U8_DCFIX):
        plan = fsea.Plan(n, mode=mode)
        plan.exec_device(d_in.ptr, nf, d_out.ptr)
        plan.synchronize()
        px = d_out.download(np.uint8, (nf, n))
        want = O.rows_mt(iq, nf, n, mode=parity.ORACLE_MODE[mode])
        for f0 in range(0, nf, 64):
            parity.check_u8(px[f0:f0 + 64], want[f0:f0 + 64])
        if mode == fsea.MODE_DB5_U8_DCFIX:
            assert np.array_equal(px[:, n // 2], px[:, n // 2 - 1])        # c/fft-batch-broad.c:113-115
        plan.close()
    d_in.free()
    d_out.free()


def test_linearity_and_bitwise_reproducibility_at_full_size():
    """Two size-independent properties on the BASELINE batch: the transform is linear
    (X[a + b] = X[a] + X[b] up to fp32 rounding; the offset-binary DC sits in bin N/2 once per
    transform), and a launch is deterministic down to the bit although frames are handed out
    dynamically (tickets, stealing), also across different launch geometries (a sub-batch)."""
    n, nf = 8192, 4096
    rng = np.random.default_rng(21)
    a = np.clip(np.rint(rng.normal(0, 14, 2 * nf * n)), -60, 60).astype(np.int8)
    b = np.clip(np.rint(rng.normal(0, 14, 2 * nf * n)), -60, 60).astype(np.int8)
    c = (a + b).astype(np.int8)                                          # |a + b| <= 120: no wrap
    plan = fsea.Plan(n, mode=fsea.MODE_COMPLEX_F32)
    d_in = DeviceBuffer(a.nbytes)
    d_out = DeviceBuffer(nf * n * 8)
    spectra = []
    for x in (a, b, c):
        d_in.upload(x.view(np.uint8))
        plan.exec_device(d_in.ptr, nf, d_out.ptr)
        plan.synchronize()
        spectra.append(d_out.download(np.complex64, (nf, n)))
    xa, xb, xc = spectra
    dc = 0.5 * n * (1 + 1j)                                              # every u8 transform carries it once
    rows = np.unique(np.r_[0, nf - 1, rng.integers(0, nf, 62)])
    want = xa[rows].astype(np.complex128) + xb[rows].astype(np.complex128)
    want[:, n // 2] -= dc
    got = xc[rows].astype(np.complex128)
    err = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
    assert err.max() < 1e-6, err.max()
    # the same launch again, and the second half alone: identical bits
    plan.exec_device(d_in.ptr, nf, d_out.ptr)
    plan.synchronize()
    again = d_out.download(np.complex64, (nf, n))
    assert np.array_equal(again.view(np.uint32), xc.view(np.uint32))
    half = nf // 2
    plan.exec_device(d_in.ptr.value + 2 * half * n, nf - half, d_out.ptr)
    plan.synchronize()
    tail = d_out.download(np.complex64, (nf, n))[: nf - half]
    assert np.array_equal(tail.view(np.uint32), xc[half:].view(np.uint32))
    d_in.free()
    d_out.free()
    plan.close()


@pytest.mark.parametrize("n,nf,mode", [(128, 33, 1), (1024, 9, 0), (2048, 5, 3), (4096, 3, 2), (8192, 13, 0),
                                       (8192, 1, 5), (16384, 2, 1)])
def test_no_write_outside_the_output_rows(n, nf, mode, units_policy):
    """Ragged launches (frame counts that do not fill the last unit, fewer frames than workgroups)
    must not touch a byte before or after their n_frames rows: guard bands around the output."""
    guard = 1 << 16
    plan = fsea.Plan(n, mode=mode)
    out_bytes = nf * plan.row_bytes
    iq = synth_iq(77 + n + nf, 2 * nf * n)
    d_in = DeviceBuffer(iq.nbytes).upload(iq)
    pattern = np.full(guard + out_bytes + guard, 0xA5, np.uint8)
    d_buf = DeviceBuffer(pattern.nbytes).upload(pattern)
    plan.exec_device(d_in.ptr, nf, ctypes.c_void_p(d_buf.ptr.value + guard))
    plan.synchronize()
    back = d_buf.download(np.uint8, (pattern.nbytes,))
    assert np.all(back[:guard] == 0xA5) and np.all(back[guard + out_bytes:] == 0xA5)
    got = back[guard: guard + out_bytes].view(plan.out_dtype).reshape(nf, n)
    parity.check_mode(got, iq, n, nf, n, True, mode)
    # the input is not padded either: the last frame ends at the last byte of the buffer
    d_in.free()
    d_buf.free()
    plan.close()


def _repeat_upload(chunk, n_chunks):
    """Device buffer holding `chunk` (host uint8) n_chunks times back to back."""
    L = fsea.hip_lib()
    buf = DeviceBuffer(chunk.nbytes * n_chunks)
    for k in range(n_chunks):
        fsea._check(L.fsea_copy_to_device(0, ctypes.c_void_p(buf.ptr.value + k * chunk.nbytes), chunk.ctypes.data,
                                          chunk.nbytes))
    return buf


def _rows_from_device(d_out, frames, n, dtype):
    L = fsea.hip_lib()
    item = np.dtype(dtype).itemsize
    out = np.empty((len(frames), n), dtype)
    for i, f in enumerate(frames):
        fsea._check(L.fsea_copy_to_host(0, out[i].ctypes.data, ctypes.c_void_p(d_out.ptr.value + int(f) * n * item), n * item))
    return out


def _compare_every_row_of_a_periodic_stream(d_out, nf, n, period, want, dtype, slab=512):
    """Rows 0..nf-1 on the device against `want` (one period of oracle rows: row f must equal want[f % period]),
    slab by slab so that neither side needs the whole array in host memory twice."""
    L = fsea.hip_lib()
    item = np.dtype(dtype).itemsize
    buf = np.empty((slab, n), dtype)
    for f0 in range(0, nf, slab):
        cnt = min(slab, nf - f0)
        fsea._check(L.fsea_copy_to_host(0, buf.ctypes.data, ctypes.c_void_p(d_out.ptr.value + f0 * n * item), cnt * n * item))
        idx = (f0 + np.arange(cnt)) % period
        ref = want[idx]
        if np.dtype(dtype) == np.uint8:
            parity.check_u8(buf[:cnt], ref)
        else:
            parity.check_float(buf[:cnt], ref)


def test_config_c5_stft_16384_at_full_size():
    """BASELINE config 5 at its full size: 2^28 samples as one stream, 16384-point frames every 8192
    samples (32767 frames, 2 GiB of f32 rows).  EVERY row against the oracle -- the stream repeats with a period of 4096
    frames, so the oracle transforms one period (the frame across the seam included) on the host's cores and all eight
    repetitions are compared with it -- Parseval on sampled rows, and the overlap itself: frames f and f + 1 share half
    their input, which a per-frame phase ramp makes visible in the complex spectrum of a pure tone (checked through the
    oracle rows instead)."""
    n, hop = 16384, 8192
    chunk = synth_iq(5, 1 << 26)                              # 32 Mi samples, repeated 8 times
    n_chunks = 8
    total_samples = n_chunks * (chunk.size // 2)
    nf = (total_samples - n) // hop + 1
    assert nf == 32767
    d_in = _repeat_upload(chunk, n_chunks)
    d_out = DeviceBuffer(nf * n * 4)
    plan = fsea.Plan(n, hop=hop, mode=fsea.MODE_MAG_NODC_F32)
    plan.exec_device(d_in.ptr, nf, d_out.ptr)
    plan.synchronize()
    per_chunk = (chunk.size // 2) // hop                      # frames per chunk period: 4096
    frames = sorted({0, 1, nf - 1, nf - 2, per_chunk - 1, per_chunk, 3 * per_chunk - 1, 17000, 29999})
    got = _rows_from_device(d_out, frames, n, np.float32).astype(np.float64)
    stream = np.concatenate([chunk, chunk])                   # enough to cut any frame, also across a chunk seam
    for row, f in zip(got, frames):
        s0 = (f * hop) % (chunk.size // 2)
        iq = stream[2 * s0: 2 * (s0 + n)]
        u = (iq ^ np.uint8(0x80)).astype(np.float64) / 256.0
        assert abs(np.sum(row * row) - n * np.sum(u * u)) / (n * np.sum(u * u)) < 2e-6
    _compare_every_row_of_a_periodic_stream(d_out, nf, n, per_chunk,
                                            O.rows_mt(stream[: 2 * ((per_chunk - 1) * hop + n)], per_chunk, n, hop=hop,
                                                      mode=O.MODE_MAG_NODC), np.float32)
    d_in.free()
    d_out.free()
    plan.close()


def test_config_c4_broad_sweep_at_full_size():
    """BASELINE config 4 on one GPU at its full size: 512 centre frequencies x 256 frames x 4096 points
    -> u8 dB tiles (DB5 + DC fix) -> stitched 256 x 2 097 152 image (c/fft-stitch-broad.c geometry).
    EVERY row of every tile against the oracle (the captures repeat with a period of 8192 frames: the oracle transforms
    one period on the host's cores, all sixteen repetitions are compared with it), and the stitch against the tile stack.
    (Periodic content: the geometry at full size, not the bench's workload.  bench.py's own config-4 job -- 512 captures seeded
    4 000 000 + f, no repetition -- is compared with the oracle row by row in tests/test_gpu_bench_jobs.py, and so is config 5's
    32767-frame stream.)"""
    n, rows, tiles = 4096, 256, 512
    chunk = synth_iq(4, 1 << 26)                              # 8192 frames of IQ, repeated 16 times = 1 GiB
    d_in = _repeat_upload(chunk, 16)
    d_px = DeviceBuffer(tiles * rows * n)
    d_img = DeviceBuffer(rows * tiles * n)
    L = fsea.hip_lib()
    zero = np.zeros(1 << 24, np.uint8)
    for off in range(0, rows * tiles * n, zero.nbytes):
        fsea._check(L.fsea_copy_to_device(0, ctypes.c_void_p(d_img.ptr.value + off), zero.ctypes.data, zero.nbytes))
    plan = fsea.Plan(n, mode=fsea.MODE_DB5_U8_DCFIX)
    plan.exec_device(d_in.ptr, tiles * rows, d_px.ptr)        # tile k = frames [k*256, (k+1)*256)
    fsea.stitch_tiles_device(d_img.ptr, d_px.ptr, tiles, 0, n, n, rows, tiles * n)
    plan.synchronize()
    rng = np.random.default_rng(44)
    per_chunk = (chunk.size // 2) // n                        # 8192 frames per chunk period
    want = O.rows_mt(chunk, per_chunk, n, mode=O.MODE_DB5_U8_DCFIX)
    _compare_every_row_of_a_periodic_stream(d_px, tiles * rows, n, per_chunk, want, np.uint8, slab=2048)
    for k in sorted({0, 1, 255, 256, 511, *rng.integers(0, tiles, 6)}):
        for y in sorted({0, rows - 1, *rng.integers(0, rows, 3)}):
            k, y = int(k), int(y)
            f = k * rows + y
            tile_row = _rows_from_device(d_px, [f], n, np.uint8)[0]
            img_row = np.empty(n, np.uint8)
            fsea._check(L.fsea_copy_to_host(0, img_row.ctypes.data,
                                            ctypes.c_void_p(d_img.ptr.value + y * tiles * n + k * n), n))
            assert np.array_equal(img_row, tile_row)          # no overlap at step = width: max with 0
    # the same image written in place by the FFT kernel (no tile stack, no second pass) is the same image
    d_img2 = DeviceBuffer(rows * tiles * n)
    plan.exec_tiled_device(d_in.ptr, tiles * rows, d_img2.ptr, rows, tiles * n, 0, rows, n)
    plan.synchronize()
    for y in (0, 1, 100, rows - 1):
        a = np.empty(tiles * n, np.uint8)
        b2 = np.empty(tiles * n, np.uint8)
        fsea._check(L.fsea_copy_to_host(0, a.ctypes.data, ctypes.c_void_p(d_img.ptr.value + y * tiles * n), a.nbytes))
        fsea._check(L.fsea_copy_to_host(0, b2.ctypes.data, ctypes.c_void_p(d_img2.ptr.value + y * tiles * n), b2.nbytes))
        assert np.array_equal(a, b2)
    d_img2.free()
    for b in (d_in, d_px, d_img):
        b.free()
    plan.close()


@pytest.mark.parametrize("n", SIZES)
def test_extreme_and_degenerate_inputs(n):
    """Full-scale and degenerate captures at every size: saturated square wave (+127 / -128 on I and Q),
    all bytes equal, a single impulse, the Nyquist alternation (which (-1)^n centring turns into DC of
    the shifted spectrum) -- float rows, complex rows and pixels against the oracle."""
    frames = []
    frames.append(np.tile(np.array([0x7f, 0x80, 0x80, 0x7f], np.uint8), n // 2))          # full-scale extremes
    frames.append(np.full(2 * n, 0x80, np.uint8))                                         # constant -128
    frames.append(np.full(2 * n, 0x7f, np.uint8))                                         # constant +127
    imp = np.zeros(2 * n, np.uint8)
    imp[0], imp[1] = 0x7f, 0x80                                                           # impulse at sample 0
    frames.append(imp)
    alt = np.zeros(2 * n, np.uint8)
    alt[0::4], alt[1::4] = 0x7f, 0x7f                                                     # + + 0 0 ... : fs/2 tone
    alt[2::4], alt[3::4] = 0x81, 0x81
    frames.append(alt)
    iq = np.concatenate(frames)
    nf = len(frames)
    for mode in (0, 3, 1, 2, 5):
        plan = fsea.Plan(n, mode=mode)
        got = plan.exec_host(iq, nf)
        parity.check_mode(got, iq, n, nf, n, True, mode)
        plan.close()
    # offset-binary convention (flip = 0) on the same bytes
    plan = fsea.Plan(n, mode=fsea.MODE_COMPLEX_F32)
    got = plan.exec_host(iq, nf, flip=False)
    parity.check_mode(got, iq, n, nf, n, False, 3)
    plan.close()


def test_host_threads_share_and_split_plans(golden):
    """SURVEY 8(b) threading: nrf_fft_* may be called from the Lua thread and from a device RX thread.
    Four host threads drive four nrf_fft objects at once, then all four push rows into ONE object
    (its mutex serialises them); every row must be the spectrum of the buffer that was pushed."""
    import threading
    L = nrf.nrf_lib()
    n, h = 1024, 64
    raws = [golden["rf_100p900_1__flipped"], golden["rf_202p500_1__flipped"],
            golden["rf_202p500_2__flipped"], golden["rf_202p500_3__flipped"]]
    keys = ["rf_100p900_1", "rf_202p500_1", "rf_202p500_2", "rf_202p500_3"]
    wants = [golden["%s__mag_%d" % (k, n)] for k in keys]
    bufs = [_nut_u8(L, r) for r in raws]
    errors = []

    def own_object(i):
        try:
            fft = L.nrf_fft_new(n, h)
            for _ in range(h):
                L.nrf_fft_process(fft, bufs[i])
            out = L.nrf_fft_get_buffer(fft)
            hist = nrf.buffer_to_numpy(L, out).reshape(h, n)
            L.nut_buffer_free(out)
            L.nrf_fft_free(fft)
            for r in (0, h // 2, h - 1):
                parity.check_float(hist[r], wants[i])
        except Exception as e:                                  # surfaced in the main thread below
            errors.append(e)

    threads = [threading.Thread(target=own_object, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors

    shared = L.nrf_fft_new(n, h)

    def shared_object(i):
        for _ in range(h // 4):
            L.nrf_fft_process(shared, bufs[i])

    threads = [threading.Thread(target=shared_object, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    out = L.nrf_fft_get_buffer(shared)
    hist = nrf.buffer_to_numpy(L, out).reshape(h, n)
    L.nut_buffer_free(out)
    L.nrf_fft_free(shared)
    counts = [0, 0, 0, 0]
    for row in hist:                                            # every row is exactly one of the four spectra
        d = [np.linalg.norm(row - w) / np.linalg.norm(w) for w in wants]
        k = int(np.argmin(d))
        assert d[k] < 1e-6, d
        counts[k] += 1
    assert counts == [h // 4] * 4, counts
    for b in bufs:
        L.nut_buffer_free(b)


# ---------------------------------------------------------------------------------------------
# the reference's five FFT scenes, as TRACES OF THE SCRIPTS THEMSELVES.  tests/golden/lua_scene_traces.json was
# written in the build container by the reference's own vendored Lua 5.3 interpreter running lua/fft.lua, fft-shifted.lua,
# fft-sea.lua, fft-sea-auto.lua and fft-sea-sick.lua unmodified under a tracing host (oracle/lua_trace.c;
# tests/golden/make_lua_traces.py): every call into the nrf_* surface with the arguments the C function receives
# (src/main.cpp:775-811 binds them one to one; nrf_fft_shift's d arrives narrowed to float, main.cpp:788), every
# ngl_texture_update with its size, the key events and the garbage collections that free buffers.  The replay makes exactly
# those calls on libfsea_nrf.so and checks every buffer that crosses the boundary against the oracle applied to the same
# bytes -- and, without a taper, against the checksums the trace recorded.  (Rounds 2-4 replayed a hand-written table of
# what the scenes were believed to do; the traces show three things it had wrong: fft-sea.lua and fft-sea-auto.lua call
# nrf_fft_shift(fft, inf) from setup(), fft-sea-auto.lua retunes by 0.01 MHz at the end of EVERY draw, and fft-sea-sick.lua
# does have a retune handler.)
# ---------------------------------------------------------------------------------------------
def _lua_traces():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "lua_scene_traces.json")) as fp:
        return json.load(fp)


LUA_SCENES = ("fft.lua", "fft-shifted.lua", "fft-sea.lua", "fft-sea-auto.lua", "fft-sea-sick.lua")
# this repository's own scene (tests/golden/scenes/, traced by the same host): two nrf_fft objects, one of which chooses and
# changes its taper through nrf_fft_set_window (include/nrf.h: the addition beside the reference's five prototypes)
OWN_LUA_SCENES = ("fft-windowed.lua",)


def _num(v):
    return float(v) if isinstance(v, str) else v                      # "inf" / "-inf" / "nan"


def _fnv1a(data):
    h = 1469598103934665603
    for b in data.tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def _texture_update(L, buf, width, height):
    """src/ngl.c:224-239 for an F64 buffer: the size check, then tex[i] = (float) data.f64[i]."""
    c = buf.contents
    assert width * height <= c.length, "ERROR ngl_texture_update: Invalid width / height"
    assert c.type == nrf.NUT_BUFFER_F64 and c.channels == 1
    size = width * height * c.channels
    return np.ctypeslib.as_array(c.data.f64, shape=(c.length,))[:size].astype(np.float32)


@pytest.fixture(params=[None, "hann"], ids=["rectangular", "hann"])
def scene_taper(request, monkeypatch):
    """NRF_FFT_WINDOW for the scene replays: unset = the reference (the trace's checksums must then be reproduced), "hann" =
    the same call sequence with the fused taper, against the windowed oracle."""
    if request.param:
        monkeypatch.setenv("NRF_FFT_WINDOW", request.param)
    else:
        monkeypatch.delenv("NRF_FFT_WINDOW", raising=False)
    return request.param


@pytest.mark.parametrize("scene", LUA_SCENES + OWN_LUA_SCENES)
def test_lua_scene_trace_replay(golden, tmp_path, scene, history_mode, scene_taper):
    import time
    traces = _lua_traces()
    own = scene in OWN_LUA_SCENES
    events = traces["own_scenes" if own else "scenes"][scene]["events"]
    L = nrf.nrf_lib()
    # the replay file the generator used: four blocks, the recorded captures' first 32 KiB, zero beyond
    blocks = []
    for key in traces["replay_blocks"]:
        blk = np.zeros(nrf.NRF_BUFFER_SIZE_BYTES, np.uint8)
        blk[: golden[key + "__raw"].size] = golden[key + "__raw"]
        blocks.append(blk)
    path = tmp_path / "replay.raw"
    np.concatenate(blocks).tofile(path)
    flipped = [O.flip_u8(b) for b in blocks]
    fnv = [_fnv1a(f) for f in flipped]

    dev = None
    ffts, shifters, buffers = {}, {}, {}              # trace id -> our objects
    n_checked = {"get_buffer": 0, "texture": 0, "shift": 0}

    def window_of(f):                                 # the taper this nrf_fft object carries right now (None = rectangular)
        name = f["taper"]
        return O.window(name, f["n"]).astype(np.float32).astype(np.float64) if name else None

    for ev in events:
        kind = ev["ev"]
        if kind == "gc":
            buf = buffers.pop(ev["buffer"], None)
            if buf is not None:
                L.nut_buffer_free(buf["ptr"])
            continue
        if kind == "frame":
            L.nrf_device_step(dev)                    # the tracer's replay device moves one block per rendered frame
            continue
        if kind != "call":
            continue
        fn = ev["fn"]
        if fn == "nrf_device_new":
            assert ev["replayed_blocks"] == len(blocks) and ev["sample_rate"] == 5000000
            dev = L.nrf_device_new(ev["freq_mhz"], str(path).encode())
            L.nrf_device_set_paused(dev, 1)
        elif fn == "nrf_device_set_frequency":
            assert L.nrf_device_set_frequency(dev, ev["freq_mhz"]) == ev["ret"]
        elif fn == "nrf_device_get_samples_buffer":
            want = flipped[ev["block"]]
            assert fnv[ev["block"]] == ev["ret"]["fnv"], "the replay file is not the generator's"
            deadline = time.time() + 2.0
            while True:                               # the replay thread ingests at 60 Hz
                samples = L.nrf_device_get_samples_buffer(dev)
                got = np.ctypeslib.as_array(samples.contents.data.u8, shape=(nrf.NRF_BUFFER_SIZE_BYTES,))
                if np.array_equal(got, want):
                    break
                L.nut_buffer_free(samples)
                assert time.time() < deadline, "replay device did not deliver block %d" % ev["block"]
                time.sleep(0.005)
            c = samples.contents
            assert (c.type, c.length, c.channels) == (ev["ret"]["type"], ev["ret"]["length"], ev["ret"]["channels"])
            buffers[ev["ret"]["id"]] = {"ptr": samples, "u8": want}
        elif fn == "nrf_fft_new":
            n, h = ev["fft_size"], ev["fft_history_size"]
            # a new block starts from NRF_FFT_WINDOW (scene_taper); nrf_fft_set_window overrides it per object
            ffts[ev["ret"]] = {"ptr": L.nrf_fft_new(n, h), "n": n, "h": h, "want": np.zeros((h, n)), "taper": scene_taper}
        elif fn == "nrf_fft_set_window":
            assert own, "the reference's scenes never call the addition"
            f = ffts[ev["fft"]]
            L.nrf_fft_set_window(f["ptr"], ev["name"].encode())
            f["taper"] = None if ev["name"] in ("rect", "none", "") else ev["name"]
            n_checked["set_window"] = n_checked.get("set_window", 0) + 1
        elif fn == "nrf_fft_process":
            f, b = ffts[ev["fft"]], buffers[ev["buffer"]]
            L.nrf_fft_process(f["ptr"], b["ptr"])
            n = f["n"]
            if "u8" in b:                             # src/nrf.c:603-606 (device buffers are offset binary already)
                row = (O.rows_windowed(b["u8"][: 2 * n], 1, n, window_of(f), flip=False) if f["taper"]
                       else O.rows(b["u8"][: 2 * n], 1, n, flip=False))[0]
            else:                                     # src/nrf.c:607-612: the shifter's F64 output
                row = O.rows_f64(b["f64"][: 2 * n], 1, n, window=window_of(f))[0]
            f["want"] = np.vstack([row[None, :], f["want"][:-1]])                 # src/nrf.c:616-617: newest row first
        elif fn == "nrf_fft_shift":
            f = ffts[ev["fft"]]
            d = _num(ev["d"])                         # what the binding hands over: (double)(float) of the Lua number
            assert d == np.float32(_num(ev["d_lua"]))
            L.nrf_fft_shift(f["ptr"], d)
            O.fft_shift(f["want"], f["n"], f["h"], d)
            n_checked["shift"] += 1
        elif fn == "nrf_fft_get_buffer":
            f = ffts[ev["fft"]]
            buf = L.nrf_fft_get_buffer(f["ptr"])
            c = buf.contents
            assert (c.type, c.length, c.channels) == (ev["ret"]["type"], ev["ret"]["length"], ev["ret"]["channels"])
            got = nrf.buffer_to_numpy(L, buf).reshape(f["h"], f["n"])
            live = np.flatnonzero(f["want"].any(axis=1))
            if live.size:
                parity.check_float(got[live], f["want"][live])
            dead = np.setdiff1d(np.arange(f["h"]), live)
            assert not got[dead].any()
            if not scene_taper:                       # the number the reference's own script saw at this point
                assert abs(float(got.sum()) - _num(ev["ret"]["sum"])) <= 2e-6 * max(1.0, _num(ev["ret"]["abs_sum"]))
            buffers[ev["ret"]["id"]] = {"ptr": buf, "hist": got}
            n_checked["get_buffer"] += 1
        elif fn == "nrf_freq_shifter_new":
            shifters[ev["ret"]] = {"ptr": L.nrf_freq_shifter_new(ev["freq_offset"], ev["sample_rate"]),
                                   "args": (ev["freq_offset"], ev["sample_rate"]), "state": (1.0, 0.0), "out": None}
        elif fn == "nrf_freq_shifter_process":
            sh, b = shifters[ev["shifter"]], buffers[ev["buffer"]]
            L.nrf_freq_shifter_process(sh["ptr"], b["ptr"])
            sh["out"], sh["state"] = O.freq_shift(b["u8"], sh["args"][0], sh["args"][1], sh["state"])
        elif fn == "nrf_freq_shifter_get_buffer":
            sh = shifters[ev["shifter"]]
            buf = L.nrf_freq_shifter_get_buffer(sh["ptr"])
            c = buf.contents
            assert (c.type, c.length, c.channels) == (ev["ret"]["type"], ev["ret"]["length"], ev["ret"]["channels"])
            vals = np.ctypeslib.as_array(c.data.f64, shape=(c.length * c.channels,)).copy()
            assert np.max(np.abs(vals[: sh["out"].size] - sh["out"])) <= 1e-9 and not vals[sh["out"].size:].any()
            assert abs(float(vals.sum()) - _num(ev["ret"]["sum"])) <= 1e-6 * _num(ev["ret"]["abs_sum"])
            buffers[ev["ret"]["id"]] = {"ptr": buf, "f64": vals}
        elif fn == "ngl_texture_update":
            b = buffers[ev["buffer"]]
            tex = _texture_update(L, b["ptr"], ev["width"], ev["height"])
            if not scene_taper:
                assert abs(float(tex.sum(dtype=np.float64)) - _num(ev["f32_sum"])) <= 2e-6 * max(1.0, abs(_num(ev["f32_sum"])))
            with pytest.raises(AssertionError, match="Invalid width / height"):   # the reference's fatal error (src/ngl.c:224-227)
                _texture_update(L, b["ptr"], ev["width"], ev["height"] + 1)
            n_checked["texture"] += 1
        else:
            raise AssertionError("trace names a call the replay does not know: %s" % fn)
    frames = sum(1 for e in events if e["ev"] == "frame")
    assert n_checked["get_buffer"] == frames * len(ffts) and n_checked["texture"] == frames and n_checked["shift"] >= 1
    assert n_checked.get("set_window", 0) == (5 if own else 0)
    for b in buffers.values():
        L.nut_buffer_free(b["ptr"])
    for sh in shifters.values():
        L.nrf_freq_shifter_free(sh["ptr"])
    for f in ffts.values():
        L.nrf_fft_free(f["ptr"])
    L.nrf_device_free(dev)


def test_launches_can_be_captured_into_a_hip_graph():
    """fsea_exec_u8_device issues nothing but a kernel launch on the caller's stream, so a sequence of launches can be
    stream-captured (torch.cuda.graph = hipGraph on ROCm) and replayed: the launch-bound form of a consumer's inner loop.
    Both frame distributions, two sizes; the replayed rows equal the directly launched ones bit for bit."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    for n, nf, policy in ((8192, 96, fsea.UNITS_AUTO), (8192, 96, fsea.UNITS_TICKETS), (1024, 500, fsea.UNITS_AUTO)):
        plan = fsea.Plan(n)
        plan.set_unit_distribution(policy)
        iqs = [torch.from_numpy(synth_iq(70 + k, 2 * nf * n).copy()).to(dev) for k in range(4)]
        outs = [torch.zeros(nf * n, dtype=torch.float32, device=dev) for _ in range(4)]
        want = []
        for k in range(4):
            plan.exec_device(iqs[k].data_ptr(), nf, outs[k].data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want = [o.clone() for o in outs]
        for o in outs:
            o.zero_()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm the stream's counter slot outside the capture
            plan.exec_device(iqs[0].data_ptr(), nf, outs[0].data_ptr(), stream=side.cuda_stream)
        torch.cuda.synchronize()
        outs[0].zero_()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for k in range(4):
                plan.exec_device(iqs[k].data_ptr(), nf, outs[k].data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        for rep in range(3):
            for o in outs:
                o.zero_()
            graph.replay()
            torch.cuda.synchronize()
            for k in range(4):
                assert torch.equal(outs[k], want[k]), (n, policy, rep, k)
        parity.check_mode(want[1].cpu().numpy().reshape(nf, n), iqs[1].cpu().numpy(), n, nf, n, True, 0)
        del graph
        plan.close()


def test_captured_streams_give_their_counter_slot_back():
    """A captured launch reserves its stream's ticket-counter slot (the graph's kernel node holds the address).  An
    application that captures on short-lived streams hands each slot back with fsea_plan_release_stream once the graph is
    gone: 80 capture / replay / release rounds on 80 distinct streams of one plan (64 slots) all work and give
    the direct rows; without the release the plan runs out at the 65th and says which call frees them.  (Streams from
    fsea_stream_create: torch.cuda.Stream() hands out 32 pooled handles round-robin.)"""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    L = fsea.hip_lib()
    L.fsea_stream_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.fsea_stream_destroy.argtypes = [ctypes.c_int, ctypes.c_void_p]

    def new_stream():
        h = ctypes.c_void_p()
        fsea._check(L.fsea_stream_create(0, ctypes.byref(h)))
        return h.value, torch.cuda.ExternalStream(h.value, device=dev)

    n, nf = 8192, 96
    iq = torch.from_numpy(synth_iq(83, 2 * nf * n).copy()).to(dev)
    out = torch.zeros(nf * n, dtype=torch.float32, device=dev)
    plan = fsea.Plan(n)
    plan.set_unit_distribution(fsea.UNITS_TICKETS)
    plan.exec_device(iq.data_ptr(), nf, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = out.clone()
    seen = set()
    for rnd in range(80):
        handle, side = new_stream()
        seen.add(handle)
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            plan.exec_device(iq.data_ptr(), nf, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want), rnd
        del graph
        plan.release_stream(handle)
        del side
        fsea._check(L.fsea_stream_destroy(0, handle))
    plan.release_stream(0)                                  # a stream without a slot: nothing to do, FSEA_OK
    plan.close()
    plan = fsea.Plan(n)
    plan.set_unit_distribution(fsea.UNITS_TICKETS)
    keep = []
    with pytest.raises(fsea.FseaError, match="fsea_plan_release_stream"):
        for rnd in range(70):
            handle, side = new_stream()
            keep.append((handle, side))
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                plan.exec_device(iq.data_ptr(), nf, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
            keep[-1] += (graph,)
    assert len(keep) == 65                                  # 64 slots held by captured streams, the 65th is refused
    torch.cuda.synchronize()
    plan.release_stream(keep[0][0])                         # one slot back: the next launch finds it
    plan.exec_device(iq.data_ptr(), nf, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    handles = [k[0] for k in keep]
    del keep, graph, side
    plan.reset()
    plan.close()
    for h in handles:
        fsea._check(L.fsea_stream_destroy(0, h))


def test_windowed_launches_capture_too_and_the_anysize_paths_refuse_a_capturing_stream():
    """A windowed plan's launch is one kernel as well (replayed rows equal the direct ones); a plan of a size without a
    kernel of its own (Bluestein: several launches through plan-owned work buffers, ordered by events) refuses a
    capturing stream with a clear FSEA_EINVAL instead of an opaque HIP error."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    n, nf = 4096, 64
    plan = fsea.Plan(n)
    plan.set_window("hann")
    iq = torch.from_numpy(synth_iq(81, 2 * nf * n).copy()).to(dev)
    out = torch.zeros(nf * n, dtype=torch.float32, device=dev)
    plan.exec_device(iq.data_ptr(), nf, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = out.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        plan.exec_device(iq.data_ptr(), nf, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    del graph
    plan.close()
    odd = fsea.Plan(1000, hop=1000)
    iq2 = torch.from_numpy(synth_iq(82, 2 * 8 * 1000).copy()).to(dev)
    out2 = torch.zeros(8 * 1000, dtype=torch.float32, device=dev)
    odd.exec_device(iq2.data_ptr(), 8, out2.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)   # works outside a capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with pytest.raises(fsea.FseaError, match="cannot be captured"):
        with torch.cuda.graph(graph, stream=side):
            odd.exec_device(iq2.data_ptr(), 8, out2.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    odd.close()


def test_streams_and_asynchronous_copies_of_the_c_abi():
    """fsea_stream_create + fsea_copy_to_device_async / fsea_copy_to_host_async: the two-stream pattern of INTEGRATION.md
    section 2 through ctypes -- batches alternate between two streams with their own device buffers, pinned host memory."""
    L = fsea.hip_lib()
    n, nf, batches = 2048, 300, 6
    plan = fsea.Plan(n, mode=fsea.MODE_DB10_U8)
    vp = ctypes.c_void_p
    L.fsea_stream_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.fsea_stream_destroy.argtypes = [ctypes.c_int, vp]
    L.fsea_copy_to_device_async.argtypes = [ctypes.c_int, vp, vp, ctypes.c_size_t, vp]
    L.fsea_copy_to_host_async.argtypes = [ctypes.c_int, vp, vp, ctypes.c_size_t, vp]
    slots = []
    for k in range(2):
        st = vp()
        fsea._check(L.fsea_stream_create(0, ctypes.byref(st)))
        slots.append((st, DeviceBuffer(2 * nf * n), DeviceBuffer(nf * n), fsea.PinnedArray((2 * nf * n,), np.uint8),
                      fsea.PinnedArray((nf, n), np.uint8)))
    iqs = [synth_iq(90 + b, 2 * nf * n) for b in range(batches)]
    got = []
    for b in range(batches + 2):
        st, d_in, d_out, h_in, h_out = slots[b % 2]
        if b >= 2:                                                    # what this slot took two batches ago
            fsea._check(L.fsea_stream_synchronize(plan._p, st))
            got.append(h_out.array.copy())
        if b < batches:
            h_in.array[:] = iqs[b]
            fsea._check(L.fsea_copy_to_device_async(0, d_in.ptr, h_in.array.ctypes.data, h_in.array.nbytes, st))
            plan.exec_device(d_in.ptr, nf, d_out.ptr, stream=st.value)
            fsea._check(L.fsea_copy_to_host_async(0, h_out.array.ctypes.data, d_out.ptr, nf * n, st))
    for b in range(batches):
        parity.check_mode(got[b], iqs[b], n, nf, n, True, fsea.MODE_DB10_U8)
    for st, d_in, d_out, h_in, h_out in slots:
        d_in.free()
        d_out.free()
        h_in.close()
        h_out.close()
        fsea._check(L.fsea_stream_destroy(0, st))
    plan.close()


@pytest.mark.parametrize("n,nf", [(8192, 1500), (4096, 2049), (16384, 700)])
def test_frame_distribution_does_not_change_a_bit(n, nf):
    """Ticket pools and static interleave are two ways of handing the same frames to the workgroups: the rows are
    identical bit for bit (every mode that has a compile-time kernel, plus the run-time-mode kernel)."""
    iq = synth_iq(5 * n + nf, 2 * nf * n)
    d_in = DeviceBuffer(iq.nbytes).upload(iq)
    for mode, dt in ((0, np.float32), (2, np.uint8), (3, np.complex64)):
        got = {}
        for policy in (fsea.UNITS_STATIC, fsea.UNITS_TICKETS):
            plan = fsea.Plan(n, mode=mode)
            plan.set_unit_distribution(policy)
            d_out = DeviceBuffer(nf * n * np.dtype(dt).itemsize)
            plan.exec_device(d_in.ptr, nf, d_out.ptr)
            plan.exec_device(d_in.ptr, nf, d_out.ptr)          # twice: the counters come back to zero
            plan.synchronize()
            got[policy] = d_out.download(dt, (nf, n))
            d_out.free()
            plan.close()
        assert np.array_equal(got[fsea.UNITS_STATIC].view(np.uint8), got[fsea.UNITS_TICKETS].view(np.uint8)), (n, mode)
    d_in.free()


@pytest.mark.parametrize("n", SIZES)
def test_identical_launches_give_identical_rows_under_load(n):
    """Long launches (every CU holds its full complement of workgroups) of the run-time-mode kernel in the f32-row modes,
    three times each: the rows must not differ between identical launches, and sampled rows must match the oracle.
    Regression test for a rare corruption found by scripts/soak.py (round 2): with `mode == DB_F32 ? log : sqrt` per
    element the compiler emitted v_log_f32 + v_sqrt_f32 + v_cndmask, and under load lanes 12-15 of every 16-lane row
    sometimes kept a stale value -- visible only at 128 and 2048 points (four adjacent bins per lane in the last pass)
    and only beyond the first unit of the first-placed workgroups, so that no short test had ever seen it."""
    nf = max(2000, (40000 * 128) // n) if n <= 2048 else 6000
    iq = synth_iq(40 + n, 2 * nf * n)
    d_in = DeviceBuffer(iq.nbytes).upload(iq)
    d_out = DeviceBuffer(nf * n * 8)
    rng = np.random.default_rng(n)
    for mode, flip in ((fsea.MODE_MAG_F32, False), (fsea.MODE_MAG_NODC_F32, True), (fsea.MODE_DB_F32, True),
                       (fsea.MODE_COMPLEX_F32, True)):                 # complex rows: 16-byte stores at most sizes
        plan = fsea.Plan(n, mode=mode)
        dt = np.complex64 if mode == fsea.MODE_COMPLEX_F32 else np.float32
        outs = []
        for rep in range(3):
            plan.exec_device(d_in.ptr, nf, d_out.ptr, flip=flip)
            plan.synchronize()
            outs.append(d_out.download(dt, (nf, n)))
        for o in outs[1:]:
            differing = np.nonzero((outs[0].view(np.uint32).reshape(nf, -1) != o.view(np.uint32).reshape(nf, -1)).any(axis=1))[0]
            assert differing.size == 0, (n, mode, differing[:8])
        rows = sorted({0, nf - 1, *rng.integers(nf // 2, nf, 5)})
        for f in rows:
            f = int(f)
            parity.check_mode(outs[0][f:f + 1], iq[2 * f * n: 2 * (f + 1) * n], n, 1, n, flip, mode)
        plan.close()
    d_in.free()
    d_out.free()


@pytest.mark.parametrize("n,nf", [(128, 40000), (2048, 6000), (4096, 3000), (16384, 700)])
def test_f64_input_rows_are_repeatable_under_load(n, nf):
    """The f32-complex kernels (NUT_BUFFER_F64 input) on launches long enough to load every CU: identical calls give
    identical rows, and sampled rows match numpy (same reason as test_identical_launches_give_identical_rows_under_load;
    these kernels hold 16-byte stores at 128 and 2048 points too)."""
    rng = np.random.default_rng(n)
    x = rng.normal(0, 0.2, 2 * nf * n)
    plan = fsea.Plan(n)
    outs = [plan.exec_host_f64(x, nf) for _ in range(3)]
    for o in outs[1:]:
        differing = np.nonzero((outs[0].view(np.uint32) != o.view(np.uint32)).any(axis=1))[0]
        assert differing.size == 0, (n, differing[:8])
    for f in sorted({0, nf - 1, *rng.integers(nf // 2, nf, 4)}):
        f = int(f)
        z = (x[2 * f * n: 2 * (f + 1) * n: 2] + 1j * x[2 * f * n + 1: 2 * (f + 1) * n: 2]) * (1.0 - 2.0 * (np.arange(n) & 1))
        want = np.abs(np.fft.fft(z))
        want[n // 2] = want[n // 2 - 1]
        assert np.linalg.norm(outs[0][f] - want) / np.linalg.norm(want) < 1e-6, (n, f)
    plan.close()
