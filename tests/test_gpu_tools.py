"""GPU tier: the sweep tools (frequensea_amd/bin/fsea-fft-batch, fsea-fft-stitch; C on the C ABI)
against the oracle's restatement of c/fft-batch*.c and c/fft-stitch*.c, PNGs decoded with PIL."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import ROOT, synth_iq

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "frequensea_amd", "bin")
TRANSFER = 262144


def _capture(path, seed, transfers, zero=False):
    raw = np.zeros(transfers * TRANSFER, np.uint8) if zero else synth_iq(seed, transfers * TRANSFER)
    raw.tofile(path)
    return raw.reshape(transfers, TRANSFER)


def _expected_tile(blocks, n, rows, skip, mode):
    """Row y (newest first) = first n samples of transfer skip + rows - 1 - y (c/fft-batch.c:62-74)."""
    return np.stack([O.rows(blocks[skip + rows - 1 - y, : 2 * n], 1, n, mode=mode)[0] for y in range(rows)])


def _png(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im)


def _close(got, want):
    d = np.abs(got.astype(int) - want.astype(int))
    return d.max() <= 1 and np.count_nonzero(d) <= max(1, got.size // 1000)


def test_broad_sweep_gate_and_stitch(tmp_path):
    n, rows, skip = 256, 120, 3
    freqs = [660, 665, 670, 675]
    caps = {f: _capture(tmp_path / ("c%d.raw" % f), f, rows + skip, zero=(f == 670)) for f in freqs}
    args = [os.path.join(BIN, "fsea-fft-batch"), "--broad", "--rows", str(rows), "--skip", str(skip),
            "--out", str(tmp_path)] + ["%d=%s" % (f, tmp_path / ("c%d.raw" % f)) for f in freqs]
    out = subprocess.run(args, capture_output=True, text=True, check=True).stdout
    assert "Not interesting. Skipping..." in out          # the all-zero capture: mean |X| = 0.71 < 1.1
    assert not os.path.exists(tmp_path / "broad-670.png")
    tiles = {}
    for f in (660, 665, 675):
        got = _png(tmp_path / ("broad-%d.png" % f))
        want = _expected_tile(caps[f], n, rows, skip, O.MODE_DB5_U8_DCFIX)
        assert got.shape == (rows, n) and _close(got, want)
        assert np.array_equal(got[:, n // 2], got[:, n // 2 - 1])
        tiles[f] = got
    # stitch 660..665 (two tiles, step 5 MHz at 5 Msps -> WIDTH_STEP = 256, no overlap)
    subprocess.run([os.path.join(BIN, "fsea-fft-stitch"), "--broad", "--start", "660", "--end", "665",
                    "--dir", str(tmp_path)], capture_output=True, text=True, check=True)
    img = _png(tmp_path / "broad-stitched-660-665.png")
    want = np.zeros((rows, 2 * n), np.uint8)
    O.composite_max(want, np.ascontiguousarray(tiles[660]), 0)
    O.composite_max(want, np.ascontiguousarray(tiles[665]), n)
    assert np.array_equal(img, want)


def test_narrow_sweep_overlapping_stitch(tmp_path):
    n, rows, skip = 1024, 33, 10
    freqs = [1802.0, 1804.0, 1806.0]
    caps = [_capture(tmp_path / ("n%d.raw" % i), 50 + i, rows + skip) for i in range(3)]
    args = [os.path.join(BIN, "fsea-fft-batch"), "--rows", str(rows), "--out", str(tmp_path)]
    args += ["%.4f=%s" % (f, tmp_path / ("n%d.raw" % i)) for i, f in enumerate(freqs)]
    subprocess.run(args, capture_output=True, text=True, check=True)
    tiles = []
    for i, f in enumerate(freqs):
        got = _png(tmp_path / ("fft-%.4f.png" % f))
        assert _close(got, _expected_tile(caps[i], n, rows, skip, O.MODE_DB10_U8))
        tiles.append(got)
    subprocess.run([os.path.join(BIN, "fsea-fft-stitch"), "--start", "1802", "--end", "1806", "--dir", str(tmp_path)],
                   capture_output=True, text=True, check=True)
    img = _png(tmp_path / "fft-stitched-1802.0000-1806.0000.png")
    step = 1024 // (5000000 // 2000000)                   # 512: 50 % overlap (c/fft-stitch.c:25)
    want = np.zeros((rows, 1024 + 2 * step), np.uint8)
    for k, t in enumerate(tiles):
        O.composite_max(want, np.ascontiguousarray(t), k * step)
    assert img.shape == want.shape and np.array_equal(img, want)

    # the same stitch with the reference's 600-row footer: spectrogram rows unchanged, ruler below
    # (c/fft-stitch.c:191-217); labels are dot-matrix glyphs without --font, everything else equals the restatement
    subprocess.run([os.path.join(BIN, "fsea-fft-stitch"), "--start", "1802", "--end", "1806", "--footer", "600",
                    "--dir", str(tmp_path)], capture_output=True, text=True, check=True)
    img = _png(tmp_path / "fft-stitched-1802.0000-1806.0000.png")
    assert img.shape == (rows + 600, want.shape[1]) and np.array_equal(img[:rows], want)
    axis, labels = O.frequency_axis(want.shape[1], rows + 600, rows, 1024, 5000000, 2000000, 1802000000, 1806000000)
    markers_y = rows + (600 // 2 - 48 // 2)
    outside = np.ones(img.shape, bool)
    outside[markers_y:markers_y + 48] = False
    assert np.array_equal(img[rows:][outside[rows:]], axis[rows:][outside[rows:]])
    assert labels[0][1] == "1800.00" and img[markers_y:markers_y + 48, labels[0][0]:].any()


def test_gpu_stitch_equals_the_reference_binary(tmp_path):
    """fsea-fft-stitch --broad (max-composite on the GPU) against the reference's own
    c/fft-stitch-broad.c compiled as is (oracle/_ref/fft-stitch-broad): same tiles in, same image out."""
    from tests.test_reference_tools import REF_STITCH_BROAD, _png_io, _read, run_reference_stitch_broad
    if not os.path.exists(REF_STITCH_BROAD):
        pytest.skip("oracle/_ref/fft-stitch-broad not built")
    L = _png_io()
    rng = np.random.default_rng(5)
    for f in (900, 905, 910, 915):
        t = rng.integers(0, 256, (4096, 256), dtype=np.uint8)
        assert L.write_gray_png(str(tmp_path / ("broad-%d.png" % f)).encode(), 256, 4096, t.ctypes.data) == 0
    ref = _read(L, run_reference_stitch_broad(tmp_path, 900, 915))
    os.rename(tmp_path / "broad-stitched-900-915.png", tmp_path / "reference.png")
    subprocess.run([os.path.join(BIN, "fsea-fft-stitch"), "--broad", "--start", "900", "--end", "915", "--rows", "4096",
                    "--dir", str(tmp_path)], capture_output=True, text=True, check=True)
    ours = _read(L, tmp_path / "broad-stitched-900-915.png")
    assert ours.shape == ref.shape == (4096, 1024) and np.array_equal(ours, ref)


def _all_devices():
    """"0-(n-1)" for the n GPUs of this box: the RCCL backend of the gather (distinct devices)."""
    from frequensea_amd import fsea
    return "0-%d" % (fsea.device_count() - 1)


@pytest.mark.parametrize("devices,chunk", [("0", 4), ("0,0", 1), ("0,0,0", 2), ("all", 1), ("all", 3)])
def test_multi_member_sweep_equals_batch_plus_stitch(tmp_path, devices, chunk):
    """fsea-fft-sweep: one host thread per member, tiles gathered chunk by chunk to member 0 and stitched
    there (RCCL between distinct GPUs; members sharing the one GPU of this box use the copy backend, same
    control path).  Its tiles and its stitched image must equal fsea-fft-batch followed by fsea-fft-stitch,
    gate included (the all-zero capture: no PNG, nothing in the image)."""
    n, rows, skip = 256, 120, 3
    freqs = [660, 665, 670, 675, 680]
    if devices == "all":                                  # a multi-GPU box: every GPU a member, tiles over RCCL / xGMI
        from frequensea_amd import fsea
        if fsea.device_count() < 2:
            pytest.skip("one GPU here: the RCCL backend needs distinct devices")
        devices = _all_devices()
        freqs = [660 + 5 * k for k in range(2 * fsea.device_count() + 3)]      # ragged: not a multiple of the member count
    for f in freqs:
        _capture(tmp_path / ("c%d.raw" % f), f, rows + skip, zero=(f == 670))
    caps = ["%d=%s" % (f, tmp_path / ("c%d.raw" % f)) for f in freqs]
    ref_dir, out_dir = tmp_path / "ref", tmp_path / "sweep"
    ref_dir.mkdir()
    out_dir.mkdir()
    subprocess.run([os.path.join(BIN, "fsea-fft-batch"), "--broad", "--rows", str(rows), "--skip", str(skip),
                    "--out", str(ref_dir)] + caps, capture_output=True, text=True, check=True)
    res = subprocess.run([os.path.join(BIN, "fsea-fft-sweep"), "--broad", "--devices", devices, "--chunk", str(chunk),
                          "--rows", str(rows), "--skip", str(skip), "--out", str(out_dir)] + caps,
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert "Not interesting. Skipping..." in res.stdout and not os.path.exists(out_dir / "broad-670.png")
    want = np.zeros((rows, n * len(freqs)), np.uint8)
    for k, f in enumerate(freqs):
        if f == 670:
            continue
        tile = _png(ref_dir / ("broad-%d.png" % f))
        assert np.array_equal(_png(out_dir / ("broad-%d.png" % f)), tile)
        O.composite_max(want, np.ascontiguousarray(tile), k * n)
    assert np.array_equal(_png(out_dir / ("broad-stitched-660-%d.png" % freqs[-1])), want)
    if "-" in devices:
        assert "Gather backend: rccl" in res.stdout


def test_multi_member_narrow_sweep_with_overlap(tmp_path):
    """1024-point tiles at 2 MHz steps overlap by half (c/fft-stitch.c:21-25): neighbouring members' tiles
    max-composite into the same columns on the root."""
    n, rows = 1024, 33
    freqs = [1802.0, 1804.0, 1806.0, 1808.0]
    for i in range(4):
        _capture(tmp_path / ("n%d.raw" % i), 50 + i, rows + 10)
    caps = ["%.4f=%s" % (f, tmp_path / ("n%d.raw" % i)) for i, f in enumerate(freqs)]
    res = subprocess.run([os.path.join(BIN, "fsea-fft-sweep"), "--devices", "0,0,0", "--chunk", "1", "--rows", str(rows),
                          "--out", str(tmp_path)] + caps, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert "Gather backend: copy" in res.stdout
    step = 512
    want = np.zeros((rows, n + 3 * step), np.uint8)
    for k, f in enumerate(freqs):
        O.composite_max(want, np.ascontiguousarray(_png(tmp_path / ("fft-%.4f.png" % f))), k * step)
    assert np.array_equal(_png(tmp_path / "fft-stitched-1802.0000-1808.0000.png"), want)


def test_windowed_sweep_tools(tmp_path):
    """fsea-fft-batch --window hann and fsea-fft-sweep --window hann: tiles equal the oracle's windowed pixel rows
    (x[j] = (-1)^j w[j] u8[j] / 256 in front of c/fft-batch.c's pixel loop), the sweep's stitched image equals the
    max-composite of the batch tool's tiles; an unknown window name is refused."""
    n, rows, skip = 1024, 40, 10
    freqs = [1802.0, 1804.0, 1806.0]
    caps = [_capture(tmp_path / ("w%d.raw" % i), 80 + i, rows + skip) for i in range(3)]
    pairs = ["%.1f=%s" % (f, tmp_path / ("w%d.raw" % i)) for i, f in enumerate(freqs)]
    (tmp_path / "batch").mkdir()
    (tmp_path / "sweep").mkdir()
    subprocess.run([os.path.join(BIN, "fsea-fft-batch"), "--rows", str(rows), "--window", "hann", "--out", str(tmp_path / "batch")] + pairs,
                   capture_output=True, text=True, check=True)
    w = O.window("hann", n).astype(np.float32).astype(np.float64)
    tiles = []
    for i, f in enumerate(freqs):
        got = _png(tmp_path / "batch" / ("fft-%.4f.png" % f))
        want = np.stack([O.rows_windowed(caps[i][skip + rows - 1 - y, : 2 * n], 1, n, w, mode=O.MODE_DB10_U8)[0] for y in range(rows)])
        assert got.shape == (rows, n) and _close(got, want)
        plain = np.stack([O.rows(caps[i][skip + rows - 1 - y, : 2 * n], 1, n, mode=O.MODE_DB10_U8)[0] for y in range(rows)])
        assert not _close(got, plain)                       # and they are not the rectangular tiles
        tiles.append(got)
    subprocess.run([os.path.join(BIN, "fsea-fft-sweep"), "--devices", "0,0", "--rows", str(rows), "--window", "hann", "--no-tiles",
                    "--out", str(tmp_path / "sweep")] + pairs, capture_output=True, text=True, check=True)
    import glob
    img = _png(glob.glob(str(tmp_path / "sweep" / "fft-stitched-*.png"))[0])
    step = n // 2                                           # 2 MHz steps at 5 Msps: WIDTH_STEP = 1024 / 2 (c/fft-stitch.c:21)
    want = np.zeros((rows, n + 2 * step), np.uint8)
    for k, t in enumerate(tiles):
        O.composite_max(want, np.ascontiguousarray(t), k * step)
    assert np.array_equal(img, want)
    r = subprocess.run([os.path.join(BIN, "fsea-fft-batch"), "--rows", str(rows), "--window", "kaiser", "--out", str(tmp_path)] + pairs,
                       capture_output=True, text=True)
    assert r.returncode != 0 and "unknown window" in r.stderr


def test_rccl_moves_bytes_on_this_box():
    """The RCCL backend of libfsea_rccl.so on a box with one GPU: a one-rank communicator and the gather's own grouped
    ncclSend / ncclRecv with the rank itself as the peer (fsea_comm_selftest_rccl) -- RCCL loads, initialises, runs its
    kernels on a non-blocking stream and the bytes arrive unchanged.  (With two or more GPUs the sweep tests above run
    the real gather.)"""
    import ctypes
    L = ctypes.CDLL(os.path.join(os.path.dirname(BIN), "libfsea_rccl.so"))
    L.fsea_comm_selftest_rccl.argtypes = [ctypes.c_int, ctypes.c_size_t]
    L.fsea_comm_last_error.restype = ctypes.c_char_p
    for nbytes in (1, 4096, 256 * 4096 + 3, 64 << 20):      # one byte, a tile row, a ragged chunk, 64 MiB
        rc = L.fsea_comm_selftest_rccl(0, nbytes)
        assert rc == 0, (nbytes, L.fsea_comm_last_error().decode())


def test_torch_nccl_backend_initialises_on_this_box(tmp_path):
    """bench.py's N > 1 runs use torch.distributed's nccl backend (= RCCL): world size 1 here, the calls the bench makes --
    init_process_group with a device id, barrier, all_reduce(MAX) of the timing pair, batch_isend_irecv with itself."""
    code = r"""
import os, torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.barrier()
t = torch.tensor([1.5, 2.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.tolist() == [1.5, 2.5]
src = torch.arange(1 << 20, dtype=torch.uint8, device="cuda")
dst = torch.zeros_like(src)
for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, src, 0), dist.P2POp(dist.irecv, dst, 0)]):
    w.wait()
torch.cuda.synchronize()
assert torch.equal(src, dst)
dist.destroy_process_group()
print("NCCL_OK")
"""
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "NCCL_OK" in res.stdout, res.stderr[-2000:]
