"""GPU tier: the sweep tools at the REFERENCE'S OWN CONSTANTS, end to end.

  broad   /root/reference/c/fft-batch-broad.c:14-22   256-pt x 4096 rows, 471 centre frequencies 660 ... 3010 MHz step 5,
                                                      10 transfers skipped after a retune, 100-row gate
          /root/reference/c/fft-stitch-broad.c:46-94  471 tiles side by side -> 120576 x 4096
  narrow  /root/reference/c/fft-batch.c:14-21         1024-pt x 16384 rows per centre frequency
          /root/reference/c/fft-stitch.c:16-27        300 tiles 1802 ... 2400 MHz step 2 at WIDTH_STEP 512 (overlapping
                                                      max), 11211 rows + 600-row footer -> 154112 x 11811

The HackRF is replaced by capture files (consecutive 262144-byte transfers, what c/rfcap.c writes).  The tools read the
first 2N bytes of each transfer only (c/fft-batch.c:62-69), so the captures are written SPARSE here: 471 x 4106 and
300 x 16394 transfers would be 0.5 TB and 1.3 TB of mostly unread bytes; the pages that are read take 8 GB and 20 GB.

Checkers: every tile row against the threaded oracle (orc_rows_mt, tests/parity.py's u8 tolerance); the broad stitched
image bit for bit against the reference's own c/fft-stitch-broad.c compiled as is (oracle/_ref/fft-stitch-broad) on the
same tiles; the narrow stitched image bit for bit against orc_composite_max over the tiles the tool read, and within the
u8 tolerance against the composite of the oracle's tiles (the reference's fft-stitch cannot be built here: its TrueType
labels need the font file and stb; the oracle's composite is the checker, as VERDICT r05 prescribes).  Stage wall times
go to gpurun_out/r06_reference_geometry_<name>.json (copied to profiles/)."""
import json
import os
import shutil
import subprocess
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from oracle import oracle as O
from tests import parity
from tests.conftest import ROOT
from tests.test_reference_tools import REF_STITCH_BROAD, _png_io, _read

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "frequensea_amd", "bin")
TRANSFER = 262144
SKIP = 10                                                # SAMPLE_BLOCKS_TO_SKIP


def _workdir(tmp_path, need_gb):
    """A directory with `need_gb` free: /dev/shm when it has the room (the sparse captures' pages then never meet a
    disk), else pytest's tmp_path; skips when neither has it or the memory limit is too small for the checkers."""
    limit = None
    try:
        text = open("/sys/fs/cgroup/memory.max").read().strip()
        limit = None if text == "max" else int(text)
    except (OSError, ValueError):
        pass
    try:
        import psutil
        avail = psutil.virtual_memory().available
        limit = avail if limit is None else min(limit, avail)
    except Exception:
        pass
    if limit is not None and limit < (need_gb + 40) * 2 ** 30:
        pytest.skip("needs about %d GB of memory (captures in /dev/shm + tiles + oracle rows), limit %.0f GB"
                    % (need_gb + 40, limit / 2 ** 30))
    for base in ("/dev/shm", str(tmp_path)):
        try:
            if shutil.disk_usage(base).free > need_gb * 2 ** 30:
                d = os.path.join(base, "fsea_refgeo_%d" % os.getpid())
                os.makedirs(d, exist_ok=True)
                return d
        except OSError:
            continue
    pytest.skip("needs %d GB of scratch space" % need_gb)


def _capture_rows(index, transfers, n):
    """What the tools read of capture `index`: the first 2N bytes of each of `transfers` transfers -- raw HackRF int8 IQ,
    uniform noise in -32 ... 31 (sigma 18.5; six bits of each byte of the generator's raw 64-bit words: fast enough for the
    10 GB the narrow sweep reads) plus one complex tone whose bin depends on the capture."""
    rng = np.random.default_rng(6000000 + index)
    words = rng.integers(0, 1 << 64, transfers * 2 * n // 8, dtype=np.uint64, endpoint=False)
    b = words.view(np.uint8)
    b &= 0x3F                                            # in place: no second and third 33 MB array per capture
    x = b.view(np.int8).reshape(transfers, 2 * n)
    x -= 32
    k = (index * 37 + 11) % n
    ph = 2 * np.pi * k * np.arange(n) / n
    tone = np.empty(2 * n, np.int8)
    tone[0::2] = np.rint(30 * np.cos(ph))
    tone[1::2] = np.rint(30 * np.sin(ph))
    x += tone
    return x.view(np.uint8)


def _write_sparse_capture(path, rows):
    """A file of the full capture size in which only the touched pages exist.  On tmpfs one strided numpy store through a
    memory map (no Python loop, releases the GIL: the captures are written by a thread pool); elsewhere pwrite at every
    transfer's offset (a numpy.memmap store allocates the whole file on ext4)."""
    transfers, row_bytes = rows.shape
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        os.ftruncate(fd, transfers * TRANSFER)
        if path.startswith("/dev/shm/"):
            mm = np.memmap(path, dtype=np.uint8, mode="r+", shape=(transfers, TRANSFER))
            mm[:, :row_bytes] = rows
            del mm
        else:
            for t in range(transfers):
                os.pwrite(fd, rows[t].data, t * TRANSFER)
    finally:
        os.close(fd)


def _make_captures(work, count, transfers, n, name):
    def one(i):
        rows = _capture_rows(i, transfers, n)
        _write_sparse_capture(os.path.join(work, name % i), rows)
        return rows
    with ThreadPoolExecutor(8) as pool:
        return list(pool.map(one, range(count)))


def _packed(rows, height):
    """Newest first: image row y = transfer SKIP + height - 1 - y (c/fft-batch.c:56-59,72-74)."""
    return np.ascontiguousarray(rows[SKIP:SKIP + height][::-1])


def _read_tiles(L, paths):
    with ThreadPoolExecutor(8) as pool:                  # ctypes releases the GIL: eight inflates at a time
        return list(pool.map(lambda p: _read(L, p), paths))


def _record(name, stages, extra):
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "r06_reference_geometry_%s.json" % name), "w") as fp:
            json.dump({"stages_s": stages, **extra}, fp, indent=1)
    except OSError:
        pass
    print("reference geometry %s: %s" % (name, json.dumps(stages)))


def test_broad_sweep_at_the_reference_constants_equals_the_reference_stitch(tmp_path):
    if not os.path.exists(REF_STITCH_BROAD):
        pytest.skip("oracle/_ref/fft-stitch-broad not built")
    n, height = 256, 4096
    freqs = list(range(660, 3011, 5))                    # FREQUENCY_START ... FREQUENCY_END, FREQUENCY_STEP 5e6
    assert len(freqs) == 471
    work = _workdir(tmp_path, 12)
    L = _png_io()
    stages = {}
    try:
        t = time.perf_counter()
        caps = _make_captures(work, len(freqs), height + SKIP, n, "cap-%03d.raw")
        stages["captures_written_sparse"] = time.perf_counter() - t
        t = time.perf_counter()
        args = [os.path.join(BIN, "fsea-fft-batch"), "--broad", "--timing", "--out", work]
        args += ["%d=%s" % (f, os.path.join(work, "cap-%03d.raw" % i)) for i, f in enumerate(freqs)]
        out = subprocess.run(args, capture_output=True, text=True, timeout=900)
        stages["fsea_fft_batch_broad_471_captures"] = time.perf_counter() - t
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        assert out.stdout.count("(Average power:") == 471 and "Not interesting" not in out.stdout
        batch_timing = [ln for ln in out.stdout.splitlines() if ln.startswith("Stages busy")]
        # the gate's figure (c/fft-batch-broad.c:81-98: mean |X| over the first 100 rows received) for a few captures
        powers = [float(ln.split(":")[1].strip(" )")) for ln in out.stdout.splitlines() if ln.startswith("(Average power:")]
        for i in (0, 235, 470):
            first100 = np.ascontiguousarray(caps[i][SKIP:SKIP + 100])
            want = float(O.mean_magnitude(O.rows(first100, 100, n, mode=O.MODE_COMPLEX)))
            assert abs(powers[i] - want) < 0.006, (i, powers[i], want)
        t = time.perf_counter()
        tiles = _read_tiles(L, [os.path.join(work, "broad-%d.png" % f) for f in freqs])
        stages["tiles_decoded"] = time.perf_counter() - t
        t = time.perf_counter()
        iq = np.concatenate([_packed(c, height) for c in caps])
        del caps
        want = O.rows_mt(iq, len(freqs) * height, n, mode=O.MODE_DB5_U8_DCFIX).reshape(len(freqs), height, n)
        del iq
        stages["oracle_rows_mt_%d_frames" % (len(freqs) * height)] = time.perf_counter() - t
        bad = 0
        for k, tile in enumerate(tiles):
            assert tile.shape == (height, n), (k, tile.shape)
            bad += parity.check_u8(tile, want[k])
            assert np.array_equal(tile[:, n // 2], tile[:, n // 2 - 1])     # the DC patch (c/fft-batch-broad.c:115-117)
        del want
        t = time.perf_counter()
        st = subprocess.run([os.path.join(BIN, "fsea-fft-stitch"), "--broad", "--start", "660", "--end", "3010", "--rows",
                             str(height), "--dir", work], capture_output=True, text=True, timeout=900)
        stages["fsea_fft_stitch_broad"] = time.perf_counter() - t
        assert st.returncode == 0, st.stdout[-2000:] + st.stderr[-2000:]
        assert "Image size: 120576 x 4096" in st.stdout
        stitched = os.path.join(work, "broad-stitched-660-3010.png")
        os.rename(stitched, os.path.join(work, "ours.png"))
        t = time.perf_counter()
        ref = subprocess.run([REF_STITCH_BROAD, "660", "3010"], cwd=work, capture_output=True, text=True, timeout=900)
        stages["reference_fft_stitch_broad_binary"] = time.perf_counter() - t
        if ref.returncode in (126, 127) or "error while loading shared libraries" in ref.stderr:
            pytest.skip("oracle/_ref/fft-stitch-broad cannot be executed here: %s" % ref.stderr.strip())
        assert ref.returncode == 0, ref.stdout[-2000:] + ref.stderr[-2000:]
        assert "Image size: 120576 x 4096" in ref.stdout
        t = time.perf_counter()
        ours, theirs = _read(L, os.path.join(work, "ours.png")), _read(L, stitched)
        stages["stitched_images_decoded"] = time.perf_counter() - t
        assert ours.shape == theirs.shape == (height, 120576)
        assert np.array_equal(ours, theirs)              # the reference's own tool on the same 471 tiles, bit for bit
        side_by_side = np.stack(tiles, axis=1).reshape(height, len(freqs) * n)   # WIDTH_STEP = 256: no overlap
        assert np.array_equal(ours, side_by_side)
        _record("broad", stages, {"tiles": 471, "fft_size": n, "rows": height, "image": [120576, 4096],
                                  "pixels_differing_from_the_oracle_by_1": int(bad), "pixels": 471 * height * n,
                                  "fsea_fft_batch_timing_line": batch_timing})
    finally:
        shutil.rmtree(work, ignore_errors=True)


def test_narrow_sweep_at_the_reference_constants_with_the_overlapping_stitch(tmp_path):
    n, height = 1024, 16384                              # c/fft-batch.c:14-15
    freqs = [1802.0 + 2.0 * k for k in range(300)]       # c/fft-stitch.c:20-22: 1802e6 ... 2400e6 step 2e6
    stitch_rows, footer, step = 11211, 600, 512          # c/fft-stitch.c:16-19,25: IMAGE_HEIGHT 11811 - FOOTER_HEIGHT; WIDTH_STEP
    width = n + (len(freqs) - 1) * step
    assert width == 154112
    work = _workdir(tmp_path, 30)
    L = _png_io()
    stages = {}
    try:
        t = time.perf_counter()
        caps = _make_captures(work, len(freqs), height + SKIP, n, "cap-%03d.raw")
        stages["captures_written_sparse"] = time.perf_counter() - t
        t = time.perf_counter()
        args = [os.path.join(BIN, "fsea-fft-batch"), "--timing", "--out", work]
        args += ["%.4f=%s" % (f, os.path.join(work, "cap-%03d.raw" % i)) for i, f in enumerate(freqs)]
        out = subprocess.run(args, capture_output=True, text=True, timeout=900)
        stages["fsea_fft_batch_300_captures"] = time.perf_counter() - t
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        batch_timing = [ln for ln in out.stdout.splitlines() if ln.startswith("Stages busy")]
        for i in range(len(freqs)):                      # the reads are done: the captures' pages go back
            os.unlink(os.path.join(work, "cap-%03d.raw" % i))
        t = time.perf_counter()
        tiles = _read_tiles(L, [os.path.join(work, "fft-%.4f.png" % f) for f in freqs])
        stages["tiles_decoded"] = time.perf_counter() - t
        t = time.perf_counter()
        bad = 0
        oracle_tiles = []
        for k in range(len(freqs)):                      # every row of every tile: 4.9 M frames
            want = O.rows_mt(_packed(caps[k], height), height, n, mode=O.MODE_DB10_U8)
            caps[k] = None
            assert tiles[k].shape == (height, n), (k, tiles[k].shape)
            bad += parity.check_u8(tiles[k], want)
            oracle_tiles.append(np.ascontiguousarray(want[:stitch_rows]))
        stages["oracle_rows_mt_%d_frames_and_compare" % (len(freqs) * height)] = time.perf_counter() - t
        t = time.perf_counter()
        st = subprocess.run([os.path.join(BIN, "fsea-fft-stitch"), "--start", "1802", "--end", "2400", "--rows", str(stitch_rows),
                             "--footer", str(footer), "--dir", work], capture_output=True, text=True, timeout=1200)
        stages["fsea_fft_stitch_300_tiles_with_footer"] = time.perf_counter() - t
        assert st.returncode == 0, st.stdout[-2000:] + st.stderr[-2000:]
        assert "Image size: 154112 x 11211" in st.stdout
        t = time.perf_counter()
        image = _read(L, os.path.join(work, "fft-stitched-1802.0000-2400.0000.png"))
        stages["stitched_image_decoded"] = time.perf_counter() - t
        assert image.shape == (stitch_rows + footer, width)
        t = time.perf_counter()
        want = np.zeros((stitch_rows, width), np.uint8)
        for k, tile in enumerate(tiles):                 # dst = max(dst, src) at x = k * WIDTH_STEP (c/fft-stitch.c:46-54,184-188)
            O.composite_max(want, np.ascontiguousarray(tile[:stitch_rows]), k * step)
        assert np.array_equal(image[:stitch_rows], want)
        want.fill(0)
        for k, tile in enumerate(oracle_tiles):
            O.composite_max(want, tile, k * step)
        stages["oracle_composites_and_compare"] = time.perf_counter() - t
        diff = image[:stitch_rows] != want
        differing = int(np.count_nonzero(diff))
        assert differing <= max(1, int(2e-3 * want.size))            # two tiles meet in every column
        assert np.abs(image[:stitch_rows][diff].astype(np.int16) - want[diff].astype(np.int16)).max(initial=0) <= 1
        del want, diff
        # the footer: banner lines and ticks equal the restatement of c/fft-stitch.c:191-217; the label band holds glyphs
        axis, labels = O.frequency_axis(width, stitch_rows + footer, stitch_rows, n, 5000000, 2000000, 1802000000, 2400000000)
        markers_y = stitch_rows + (footer // 2 - 48 // 2)
        below = image[stitch_rows:]
        keep = np.ones(below.shape[0], bool)
        keep[markers_y - stitch_rows: markers_y - stitch_rows + 48] = False
        assert np.array_equal(below[keep], axis[stitch_rows:][keep])
        assert labels[0][1] == "1800.00" and labels[-1][1] == "2401.00" and len(labels) == 602
        assert below[markers_y - stitch_rows: markers_y - stitch_rows + 48, labels[0][0]:].any()
        _record("narrow", stages, {"tiles": 300, "fft_size": n, "rows": height, "image": [width, stitch_rows + footer],
                                   "pixels_differing_from_the_oracle_by_1_in_the_tiles": int(bad), "pixels": 300 * height * n,
                                   "stitched_pixels_differing_from_the_oracle_composite_by_1": differing,
                                   "fsea_fft_batch_timing_line": batch_timing})
    finally:
        shutil.rmtree(work, ignore_errors=True)
