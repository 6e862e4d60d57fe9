"""GPU tier: bench.py's multi-GPU leg works on two fixed jobs -- BASELINE config 4 (the fft-batch-broad sweep: 512
captures seeded 4 000 000 + f, /root/reference/c/fft-batch-broad.c:176-206, stitched as c/fft-stitch-broad.c:62-87) and
config 5 (one 32767-frame 16384-point stream at hop 8192) -- and prints a checksum of the stitched image / of all rows as
they sit on rank 0 after the gather.  Here THOSE jobs (bench.run_broad / bench.run_stft_stream, their own code path at world
size 1, their own captures) are compared with the oracle, every row (tests/parity.py's tolerances); only then are their
checksums compared with the constants committed in tests/golden/bench_job_checksums.json -- the same constants bench.py
reads for `*_gathered_checksum_matches_pinned`.  A SCALE record with N > 1 whose line says `true` therefore carries oracle
parity of what crossed xGMI, not a GPU-vs-GPU comparison (What's weak 6 / 7, next-round item 3).

tests/golden/make_bench_checksums.py regenerates the constants (on a GPU box) through verify_jobs() below: it cannot write
a checksum the oracle comparison has not passed.  The captures come from torch's device generator; their own checksums
are pinned too, so that a torch release that changes the generator's stream shows up as "the inputs changed", not as a
parity failure."""
import argparse
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import parity
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(ROOT, "tests", "golden", "bench_job_checksums.json")


def _args(**kw):
    base = dict(regime="resident", chunks=8, no_fused_stitch=False, prewarm=0.0, no_extra=True, window=None,
                stream_frames=32767)
    base.update(kw)
    return argparse.Namespace(**base)


def verify_jobs(torch):
    """Runs both jobs through bench.py's own functions, compares every row with the oracle, returns the checksums."""
    import bench
    from frequensea_amd import sweep
    torch.cuda.set_device(0)
    sums = {}
    # ---- config 4: 512 centre frequencies x 256 frames x 4096 points -> one 256 x 2 097 152 u8 image -------------------
    keep = {}
    line = bench.run_broad(_args(), 0, 1, None, torch, 2, 1, keep=keep)
    n, rows, tiles = keep["n"], keep["rows"], keep["tiles"]
    assert (n, rows, tiles) == (4096, 256, 512) and line["gathered_checksum_ok"] is True
    sums["broad_sweep_captures"] = "%016x" % sweep.checksum(torch, keep["iq"])
    iq = keep["iq"].cpu().numpy().view(np.uint8)
    image = keep["image"].cpu().numpy()
    assert image.shape == (rows, tiles * n)
    want = O.rows_mt(iq, tiles * rows, n, mode=O.MODE_DB5_U8_DCFIX).reshape(tiles, rows, n)
    differing = 0
    for a in range(0, tiles, 32):                          # tile f = image[:, f * n : (f + 1) * n], rows = frames f * 256 ...
        got = image[:, a * n:(a + 32) * n].reshape(rows, 32, n).transpose(1, 0, 2)
        differing += parity.check_u8(got, want[a:a + 32])
    sums["broad_sweep_image"] = line["gathered_checksum"]
    sums["broad_sweep_pixels_differing_from_the_oracle_by_1"] = int(differing)
    # the ingest regime (captures from pinned host memory, chunked, two streams) must give the same image
    keep2 = {}
    line2 = bench.run_broad(_args(regime="ingest"), 0, 1, None, torch, 2, 1, keep=keep2)
    assert line2["gathered_checksum"] == line["gathered_checksum"] and torch.equal(keep2["image"], keep["image"])
    del keep, keep2, image, want, iq
    torch.cuda.empty_cache()
    # ---- config 5: one stream, 32767 frames of 16384 points at hop 8192 -> f32 magnitude rows --------------------------
    keep = {}
    line = bench.run_stft_stream(_args(), 0, 1, None, torch, 1, 1, keep=keep)
    n, hop, frames = keep["n"], keep["hop"], keep["frames"]
    assert (n, hop, frames) == (16384, 8192, 32767) and keep["s_lo"] == 0 and line["gathered_checksum_ok"] is True
    sums["stft_stream_samples"] = "%016x" % sweep.checksum(torch, keep["iq"])
    iq = keep["iq"].cpu().numpy().view(np.uint8)
    worst_rel = 0.0
    for a in range(0, frames, 4096):                       # 4096 rows at a time: 512 MiB of f64 oracle rows per slab
        b = min(frames, a + 4096)
        want = O.rows_mt(iq[2 * a * hop: 2 * ((b - 1) * hop + n)], b - a, n, hop=hop)
        rel, _ = parity.check_float(keep["rows"][a:b].cpu().numpy(), want)
        worst_rel = max(worst_rel, rel)
    sums["stft_stream_rows"] = line["gathered_checksum"]
    sums["stft_stream_worst_slab_rel_l2_vs_oracle"] = worst_rel
    return sums


def test_the_bench_jobs_are_oracle_verified_and_their_checksums_pinned():
    torch = pytest.importorskip("torch")
    import bench
    sums = verify_jobs(torch)
    with open(GOLDEN) as fp:
        pinned = json.load(fp)
    assert bench.pinned_checksums() == pinned
    for key in ("broad_sweep_captures", "stft_stream_samples"):
        assert sums[key] == pinned[key], ("the job's INPUTS changed (torch's device generator?): regenerate with "
                                          "tests/golden/make_bench_checksums.py", key, sums[key], pinned[key])
    for key in ("broad_sweep_image", "stft_stream_rows"):
        assert sums[key] == pinned[key], ("the rows passed the oracle comparison but their bits changed (a kernel's arithmetic "
                                          "changed?): regenerate with tests/golden/make_bench_checksums.py", key, sums[key], pinned[key])
    # and the line says so
    line = bench.run_broad(_args(), 0, 1, None, torch, 1, 1)
    assert line["gathered_checksum_matches_pinned"] is True and line["gathered_checksum_pinned"] == pinned["broad_sweep_image"]
    short = bench.run_stft_stream(_args(stream_frames=4095), 0, 1, None, torch, 1, 1)
    assert short["gathered_checksum_matches_pinned"] is None        # no constant for another stream length
