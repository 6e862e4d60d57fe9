"""CPU tier: the N>1 path with world_size 2 over gloo.  The orchestration (frame/tile sharding, the
gather to rank 0, the stitch offsets) is the product code of frequensea_amd/sweep.py; the per-tile
spectra come from the oracle here because there is no GPU in this tier."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from frequensea_amd import sweep
from oracle import oracle as O
from tests.conftest import synth_iq

N, H, TILES, STEP = 256, 12, 5, 128          # 50 % overlap as c/fft-stitch.c (WIDTH_STEP = N/2)


def test_partition_is_contiguous_and_balanced():
    for n_items in (0, 1, 5, 8, 471, 512):
        for world in (1, 2, 3, 8):
            ranges = [sweep.partition(n_items, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n_items
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
    assert sweep.stitched_width(1024, 300, 512) == 154112      # c/fft-stitch.c: 1024 + 299*512


def test_chunk_ranges_cover_the_range_in_order():
    for lo, hi in ((0, 0), (3, 4), (0, 5), (10, 74), (7, 1000)):
        for n_chunks in (1, 2, 8, 64):
            ch = sweep.chunk_ranges(lo, hi, n_chunks)
            if hi == lo:
                assert ch == []
                continue
            assert ch[0][0] == lo and ch[-1][1] == hi and len(ch) <= n_chunks
            assert all(a < b for a, b in ch) and all(ch[i][1] == ch[i + 1][0] for i in range(len(ch) - 1))
    # a rank's frames and the halo it re-reads (BASELINE config 5: N = 16384, hop = 8192)
    assert sweep.frame_sample_range(0, 4096, 16384, 8192) == (0, 4095 * 8192 + 16384)
    assert sweep.frame_sample_range(4096, 8191, 16384, 8192)[0] == 4096 * 8192        # starts inside rank 0's halo


def _tile(freq_index):
    iq = synth_iq(4000000 + freq_index, 2 * N * H)            # seed = 4e6 + f
    return O.rows(iq, H, N, mode=O.MODE_DB5_U8_DCFIX)


def _expected():
    img = np.zeros((H, sweep.stitched_width(N, TILES, STEP)), np.uint8)
    for f in range(TILES):
        O.composite_max(img, _tile(f), f * STEP)
    return img


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def make_tiles(lo, hi):
            return torch.from_numpy(np.stack([_tile(f) for f in range(lo, hi)]) if hi > lo
                                    else np.zeros((0, H, N), np.uint8))

        def composite(image, tile, x):
            image[:, x:x + N] = torch.maximum(image[:, x:x + N], tile)

        img = sweep.run_sweep(TILES, (H, N), make_tiles, composite, dist=dist, torch=torch, device="cpu",
                              width_step=STEP, n_chunks=2)      # chunked: ranks with 2 tiles send them one by one
        # the bench's timing reduction: max over ranks
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t[0]) == world
        if rank == 0:
            np.save(out_path, img.numpy())
        else:
            assert img is None
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world", [2, 3])
def test_sweep_gather_and_stitch_over_gloo(tmp_path, world):
    out = str(tmp_path / "stitched.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert np.array_equal(np.load(out), _expected())


def _disjoint_worker(rank, world, port, out_path, step):
    """The fft-batch-broad shape (tiles side by side, WIDTH_STEP >= N): rank 0 writes its own tiles in place."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def make_tiles(lo, hi):
            assert rank != 0, "rank 0 has write_tiles"
            return torch.from_numpy(np.stack([_tile(f) for f in range(lo, hi)]))

        def write_tiles(image, lo, hi):
            calls.append((lo, hi))
            for f in range(lo, hi):
                image[:, f * step: f * step + N] = torch.from_numpy(_tile(f))

        def composite(image, tile, x):
            raise AssertionError("disjoint tiles are copied, not composited")

        img = sweep.run_sweep(TILES, (H, N), make_tiles, composite, dist=dist, torch=torch, device="cpu",
                              width_step=step, n_chunks=2, write_tiles=write_tiles)
        if rank == 0:
            assert calls and calls[0][0] == 0
            np.save(out_path, img.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,step", [(2, N), (3, N), (2, N + 24)])
def test_disjoint_sweep_written_in_place_over_gloo(tmp_path, world, step):
    out = str(tmp_path / "stitched.npy")
    mp.spawn(_disjoint_worker, args=(world, _free_port(), out, step), nprocs=world, join=True)
    want = np.zeros((H, sweep.stitched_width(N, TILES, step)), np.uint8)
    for f in range(TILES):
        O.composite_max(want, _tile(f), f * step)
    assert np.array_equal(np.load(out), want)


def test_overlapping_sweep_ignores_write_tiles():
    """write_tiles is only valid for disjoint tiles: with WIDTH_STEP < N the max-composite path runs."""
    def make_tiles(lo, hi):
        return torch.from_numpy(np.stack([_tile(f) for f in range(lo, hi)]))

    def composite(image, tile, x):
        image[:, x:x + N] = torch.maximum(image[:, x:x + N], tile)

    def write_tiles(image, lo, hi):
        raise AssertionError("not for overlapping tiles")

    img = sweep.run_sweep(TILES, (H, N), make_tiles, composite, dist=None, torch=torch, device="cpu", width_step=STEP,
                          write_tiles=write_tiles)
    assert np.array_equal(img.numpy(), _expected())


# ---- BASELINE config 5: one overlapped stream, frames sharded with a halo, rows gathered to rank 0 ----
SN, SHOP, SFRAMES = 512, 256, 23                    # 50 % overlap, a frame count no world size divides


def _stream():
    return synth_iq(5, 2 * ((SFRAMES - 1) * SHOP + SN))            # seed 5


def _stft_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stream = _stream()
        f_lo, f_hi = sweep.partition(SFRAMES, world, rank)
        s_lo, s_hi = sweep.frame_sample_range(f_lo, f_hi, SN, SHOP)
        shard = stream[2 * s_lo: 2 * s_hi].copy()                   # this rank's samples incl. the halo; nothing else

        def make_rows(a, b):                                        # frames a..b-1 of the global stream
            off = 2 * (a * SHOP - s_lo)
            rows = O.rows(shard[off: off + 2 * ((b - a - 1) * SHOP + SN)], b - a, SN, hop=SHOP)
            return torch.from_numpy(rows.astype(np.float32))

        out = torch.zeros((SFRAMES, SN), dtype=torch.float32) if rank == 0 else None
        sent = sweep.run_stft(SFRAMES, SN, make_rows, out, dist=dist, torch=torch, device="cpu", n_chunks=3)
        if rank == 0:
            assert sent == 0
            np.save(out_path, out.numpy())
        else:
            assert sent == (f_hi - f_lo) * SN * 4
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_stft_stream_sharded_with_halo_equals_single_process(tmp_path, world):
    """Every rank holds only its own samples plus the N - hop halo; the gathered rows must equal the
    single-process transform of the whole stream bit for bit."""
    out = str(tmp_path / "rows.npy")
    mp.spawn(_stft_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    want = O.rows(_stream(), SFRAMES, SN, hop=SHOP).astype(np.float32)
    assert np.array_equal(np.load(out), want)


@pytest.mark.parametrize("n_chunks", [1, 3, 8])
def test_single_process_chunked_sweep_matches(n_chunks):
    def make_tiles(lo, hi):
        return torch.from_numpy(np.stack([_tile(f) for f in range(lo, hi)]))

    def composite(image, tile, x):
        image[:, x:x + N] = torch.maximum(image[:, x:x + N], tile)

    img = sweep.run_sweep(TILES, (H, N), make_tiles, composite, dist=None, torch=torch, device="cpu", width_step=STEP,
                          n_chunks=n_chunks)
    assert np.array_equal(img.numpy(), _expected())


def test_single_process_sweep_matches():
    def make_tiles(lo, hi):
        return torch.from_numpy(np.stack([_tile(f) for f in range(lo, hi)]))

    def composite(image, tile, x):
        image[:, x:x + N] = torch.maximum(image[:, x:x + N], tile)

    img = sweep.run_sweep(TILES, (H, N), make_tiles, composite, dist=None, torch=torch, device="cpu", width_step=STEP)
    assert np.array_equal(img.numpy(), _expected())


# ---- bench.py's multi-GPU leg: what arrived against what was sent -------------------------------------------------
def test_checksum_sees_position_and_value():
    a = torch.arange(1000, dtype=torch.uint8).reshape(10, 100)
    base = sweep.checksum(torch, a)
    assert base == sweep.checksum(torch, a.clone()) and 0 <= base < 2 ** 64
    b = a.clone()
    b[3, 7] ^= 1
    assert sweep.checksum(torch, b) != base                     # one bit
    c = a.clone()
    c[0], c[1] = a[1], a[0]
    assert sweep.checksum(torch, c) != base                     # two rows swapped: a plain sum would not notice
    assert sweep.checksum(torch, a.reshape(-1)[:999]) != base   # a tail shorter than a word is zero-padded, not dropped
    f = torch.linspace(0, 1, 4099, dtype=torch.float32)
    assert sweep.checksum(torch, f) == sweep.checksum(torch, f.view(torch.int32))      # bytes, whatever the dtype
    # layout independence the leg relies on: tiles side by side in an image == the owner's [cnt, H, N] stack
    stack = torch.randint(0, 256, (3, 4, 8), dtype=torch.uint8)
    image = torch.zeros((4, 5 * 8), dtype=torch.uint8)
    image[:, 8:32].view(4, 3, 8).copy_(stack.permute(1, 0, 2))
    assert torch.equal(sweep.tiles_of_image(image, 1, 4, 8), stack)
    assert sweep.checksum(torch, sweep.tiles_of_image(image, 1, 4, 8)) == sweep.checksum(torch, stack)


def _leg_worker(rank, world, port, out_path, corrupt):
    """The verification of bench.py's multi-GPU leg, with the oracle as the source: every rank's checksum of its own tiles
    (all_gather_object) against rank 0's checksum of the same tiles as they sit in the gathered image."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = sweep.partition(TILES, world, rank)
        mine = torch.from_numpy(np.stack([_tile(f) for f in range(lo, hi)]))

        def make_tiles(a, b):
            t = mine[a - lo: b - lo].clone()
            if corrupt and rank == world - 1 and a == lo:
                t[0, 0, 0] ^= 0x40                               # a byte damaged on its way
            return t

        def write_tiles(image, a, b):
            image[:, a * N: b * N].view(H, b - a, N).copy_(mine[a - lo: b - lo].permute(1, 0, 2))

        img = sweep.run_sweep(TILES, (H, N), make_tiles, None, dist=dist, torch=torch, device="cpu", n_chunks=2,
                              write_tiles=write_tiles)
        sums = [None] * world
        dist.all_gather_object(sums, (lo, hi, sweep.checksum(torch, mine)))
        if rank == 0:
            arrived = [sweep.checksum(torch, sweep.tiles_of_image(img, a, b, N)) for a, b, _ in sums]
            np.save(out_path, np.array([x == y for x, (_, _, y) in zip(arrived, sums)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,corrupt", [(2, False), (3, False), (3, True)])
def test_gathered_checksums_match_what_the_members_computed(tmp_path, world, corrupt):
    out = str(tmp_path / "ok.npy")
    mp.spawn(_leg_worker, args=(world, _free_port(), out, corrupt), nprocs=world, join=True)
    ok = np.load(out)
    assert ok.tolist() == [True] * (world - 1) + [not corrupt]
