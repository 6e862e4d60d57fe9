"""Multi-GPU orchestration of the fft-batch / fft-stitch sweep.

The reference sweeps centre frequencies one after another on one HackRF and one CPU
(c/fft-batch-broad.c:176-206) and stitches the per-frequency tiles afterwards
(c/fft-stitch-broad.c:62-87).  Tiles are independent, so here rank r of a one-process-per-GPU job
takes a contiguous range of centre frequencies, turns each capture into a u8 dB tile on its own GPU
(no data-path collective), and only the finished tiles travel: one gather to rank 0 (RCCL over xGMI
when the backend is "nccl") followed by the max-composite of c/fft-stitch*.c:46-54 on rank 0.

The per-tile compute and the composite are passed in as callables so that the same orchestration
runs on GPUs (frequensea_amd.fsea kernels, see bench.py) and in the world_size-2 gloo test on CPU
(where the test supplies the oracle as the tile source).  Nothing here computes spectra itself.
"""
import numpy as np


def partition(n_items, world, rank):
    """Contiguous, balanced range [lo, hi) of items for `rank` (first n_items % world ranks get one more)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def stitched_width(fft_size, n_tiles, width_step):
    """IMAGE_WIDTH = FFT_SIZE + (n_tiles - 1) * WIDTH_STEP (c/fft-stitch.c:27, fft-stitch-broad.c:56)."""
    return fft_size + (n_tiles - 1) * width_step if n_tiles else 0


def run_sweep(n_tiles, tile_shape, make_tiles, composite, dist=None, torch=None, device=None, width_step=None,
              composite_stack=None):
    """Shard `n_tiles` centre frequencies over the ranks of `dist`, gather the tiles to rank 0 and stitch.

    make_tiles(lo, hi) -> torch.uint8 tensor [hi-lo, H, N] on `device` (this rank's tiles, in order)
    composite(image, tile, x)   max-composites one [H, N] tile into image[:, x:x+N] (rank 0 only)
    composite_stack(image, stack, count, first_x)   optional: all `count` tiles of one rank's stack at
                                once (tile k at first_x + k * width_step); used instead of `composite`
    Returns the stitched [H, W] torch.uint8 image on rank 0, None elsewhere.
    """
    h, n = tile_shape
    width_step = n if width_step is None else width_step
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    lo, hi = partition(n_tiles, world, rank)
    mine = make_tiles(lo, hi)
    assert tuple(mine.shape) == (hi - lo, h, n) and mine.dtype == torch.uint8
    per_rank = [partition(n_tiles, world, r) for r in range(world)]
    cap = max(b - a for a, b in per_rank)
    if world > 1:
        # equal-sized messages: pad the short ranks' stacks by one empty tile
        send = mine
        if hi - lo < cap:
            send = torch.zeros((cap, h, n), dtype=torch.uint8, device=device)
            send[: hi - lo] = mine
        gathered = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
        dist.gather(send.contiguous(), gather_list=gathered, dst=0)
    else:
        gathered = [mine]
    if rank != 0:
        return None
    image = torch.zeros((h, stitched_width(n, n_tiles, width_step)), dtype=torch.uint8, device=device)
    for r, (a, b) in enumerate(per_rank):
        if composite_stack is not None:
            composite_stack(image, gathered[r], b - a, a * width_step)
            continue
        for k in range(b - a):
            composite(image, gathered[r][k], (a + k) * width_step)
    return image


def numpy_tiles_to_torch(torch, tiles, device):
    return torch.from_numpy(np.ascontiguousarray(tiles)).to(device)
