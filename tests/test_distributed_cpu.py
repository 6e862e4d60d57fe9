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
                              width_step=STEP)
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


def test_single_process_sweep_matches():
    def make_tiles(lo, hi):
        return torch.from_numpy(np.stack([_tile(f) for f in range(lo, hi)]))

    def composite(image, tile, x):
        image[:, x:x + N] = torch.maximum(image[:, x:x + N], tile)

    img = sweep.run_sweep(TILES, (H, N), make_tiles, composite, dist=None, torch=torch, device="cpu", width_step=STEP)
    assert np.array_equal(img.numpy(), _expected())
