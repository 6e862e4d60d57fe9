"""Multi-GPU orchestration of the fft-batch / fft-stitch sweep and of the sharded STFT stream
(SURVEY.md 8(e)).

The reference sweeps centre frequencies one after another on one HackRF and one CPU
(c/fft-batch-broad.c:176-206) and stitches the per-frequency tiles afterwards
(c/fft-stitch-broad.c:62-87).  Tiles are independent, so here rank r of a one-process-per-GPU job
takes a contiguous range of centre frequencies, turns each capture into a u8 dB tile on its own GPU
(no data-path collective), and only the finished tiles travel to rank 0, where they are
max-composited (c/fft-stitch*.c:46-54).  The same holds for whole frames of one long capture
(BASELINE config 5: 16384-point, 50 % overlap): rank r transforms a contiguous frame range and reads
the N - hop samples it shares with its neighbour redundantly from the source -- no exchange -- and the
rows travel to rank 0.

The exchange is a CHUNKED gather: every rank cuts its range into chunks, computes chunk j + 1 while
chunk j is on the wire, and rank 0 posts the receives of one chunk index from all peers as one group
(torch.distributed.batch_isend_irecv = one ncclGroup of ncclSend/ncclRecv when the backend is "nccl",
i.e. RCCL: xGMI is point-to-point, every peer -> root stream rides its own link) and
consumes chunk j while chunk j + 1 arrives.  ProcessGroupNCCL runs the transfers on its own stream
behind the compute stream's work at the time of the call, so transfer and compute overlap without
further plumbing; with "gloo" (the CPU tests) the same calls are plain TCP sends.

The per-item compute and the consumer are passed in as callables so that the same orchestration
runs on GPUs (frequensea_amd.fsea kernels, see bench.py) and in the world_size-2/3 gloo tests on
CPU (where the tests supply the oracle as the source).  Nothing here computes spectra itself.
"""
import numpy as np


def partition(n_items, world, rank):
    """Contiguous, balanced range [lo, hi) of items for `rank` (first n_items % world ranks get one more)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def chunk_ranges(lo, hi, n_chunks):
    """[lo, hi) cut into at most n_chunks contiguous, balanced, non-empty ranges (empty list for an empty range)."""
    count = hi - lo
    n = max(1, min(n_chunks, count)) if count > 0 else 0
    return [(lo + partition(count, n, j)[0], lo + partition(count, n, j)[1]) for j in range(n)]


def stitched_width(fft_size, n_tiles, width_step):
    """IMAGE_WIDTH = FFT_SIZE + (n_tiles - 1) * WIDTH_STEP (c/fft-stitch.c:27, fft-stitch-broad.c:56)."""
    return fft_size + (n_tiles - 1) * width_step if n_tiles else 0


def frame_sample_range(f_lo, f_hi, n, hop):
    """Samples [s_lo, s_hi) that frames [f_lo, f_hi) of an overlapped stream read: the last N - hop of them are
    the halo shared with the next rank's first frame (read redundantly, not exchanged)."""
    if f_hi <= f_lo:
        return f_lo * hop, f_lo * hop
    return f_lo * hop, (f_hi - 1) * hop + n


def gather_chunked(n_items, item_shape, dtype, produce, consume, dist=None, torch=None, device=None, n_chunks=8):
    """Items [0, n_items) are partitioned over the ranks; rank r computes its range chunk by chunk with
    produce(a, b) -> tensor [b - a, *item_shape] of `dtype` on `device`, and rank 0 receives every chunk
    and calls consume(a, b, tensor) for it (own chunks included), chunk index by chunk index.  On rank 0
    produce may return None: "these items are already where consume would put them" (written in place).

    Returns the number of bytes this rank put on the wire (0 on rank 0)."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    ranges = [chunk_ranges(*partition(n_items, world, r), n_chunks) for r in range(world)]
    mine = ranges[rank]
    # gloo moves host memory only: device tensors are staged through the host on both sides (the
    # single-GPU exercise of the N > 1 control path, FSEA_BENCH_BACKEND=gloo); RCCL takes them as they are
    staged = dist is not None and dist.get_backend() != "nccl" and device is not None and str(device) != "cpu"
    wire = "cpu" if staged else device
    if rank != 0:
        pending, sent = [], 0
        for a, b in mine:
            t = produce(a, b).contiguous()
            assert tuple(t.shape) == (b - a,) + tuple(item_shape) and t.dtype == dtype
            if staged:
                t = t.cpu()
            # a one-op batch, not a bare isend: ProcessGroupNCCL routes batched point-to-point operations through the job's
            # communicator and bare ones through a two-rank communicator of their own -- the root's receives are batched
            # (one ncclGroup per chunk index), so the members' sends must be too, or the two sides sit on different communicators
            reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, t, 0)])
            pending.append((reqs, t))                      # the tensor stays alive until its send has completed
            sent += t.numel() * t.element_size()
        for reqs, _ in pending:
            for req in reqs:
                req.wait()
        return sent
    # rank 0: receives of chunk index j from all peers are one group; its own chunk j is computed meanwhile
    depth = max(len(r) for r in ranges) if ranges else 0
    inbox = []
    for j in range(depth):
        ops, bufs = [], []
        for r in range(1, world):
            if j < len(ranges[r]):
                a, b = ranges[r][j]
                buf = torch.empty((b - a,) + tuple(item_shape), dtype=dtype, device=wire)
                bufs.append((a, b, buf))
                ops.append(dist.P2POp(dist.irecv, buf, r))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        inbox.append((reqs, bufs))
    for j in range(depth):
        if j < len(mine):
            a, b = mine[j]
            t = produce(a, b)
            if t is not None:
                assert tuple(t.shape) == (b - a,) + tuple(item_shape) and t.dtype == dtype
                consume(a, b, t)
        reqs, bufs = inbox[j]
        for req in reqs:
            req.wait()
        for a, b, buf in bufs:
            consume(a, b, buf.to(device) if staged else buf)
    return 0


def run_sweep(n_tiles, tile_shape, make_tiles, composite, dist=None, torch=None, device=None, width_step=None,
              composite_stack=None, n_chunks=8, write_tiles=None):
    """Shard `n_tiles` centre frequencies over the ranks of `dist`, gather the tiles to rank 0 chunk by chunk
    and stitch them as they arrive.

    make_tiles(lo, hi) -> torch.uint8 tensor [hi-lo, H, N] on `device` (tiles lo..hi-1, in order)
    composite(image, tile, x)   max-composites one [H, N] tile into image[:, x:x+N] (rank 0 only)
    composite_stack(image, stack, count, first_x)   optional: all `count` tiles of a stack at once (tile k
                                at first_x + k * width_step); used instead of `composite`
    write_tiles(image, lo, hi)  optional, for sweeps whose tiles do not overlap (width_step >= N): rank 0 computes
                                its own tiles lo..hi-1 straight into the image (fsea_exec_u8_tiled_device) instead
                                of make_tiles + composite; received stacks are then copied into place (on the
                                zeroed image max(0, tile) = tile, c/fft-stitch-broad.c:62-87)
    Returns the stitched [H, W] torch.uint8 image on rank 0, None elsewhere.
    """
    h, n = tile_shape
    width_step = n if width_step is None else width_step
    rank = dist.get_rank() if dist is not None else 0
    disjoint = write_tiles is not None and width_step >= n
    image = None
    if rank == 0:
        shape = (h, stitched_width(n, n_tiles, width_step))
        # disjoint tiles that cover every column: every byte is written exactly once, no zero fill needed
        image = (torch.empty if (disjoint and width_step == n) else torch.zeros)(shape, dtype=torch.uint8, device=device)

    def produce(a, b):
        if disjoint and rank == 0:
            write_tiles(image, a, b)
            return None
        return make_tiles(a, b)

    def consume(a, b, stack):
        if disjoint:
            if width_step == n:                              # [cnt, H, N] -> image[:, a*N : b*N] seen as [H, cnt, N]
                image[:, a * n: b * n].view(h, b - a, n).copy_(stack.permute(1, 0, 2))
            else:
                for k in range(b - a):
                    image[:, (a + k) * width_step: (a + k) * width_step + n].copy_(stack[k])
            return
        if composite_stack is not None:
            composite_stack(image, stack, b - a, a * width_step)
            return
        for k in range(b - a):
            composite(image, stack[k], (a + k) * width_step)

    gather_chunked(n_tiles, (h, n), torch.uint8, produce, consume, dist=dist, torch=torch, device=device,
                   n_chunks=n_chunks)
    return image


def run_stft(n_frames, n, make_rows, out_rows, dist=None, torch=None, device=None, n_chunks=8, dtype=None):
    """BASELINE config 5: frames [0, n_frames) of ONE overlapped stream are cut into per-rank contiguous
    ranges (each rank reads its samples plus the N - hop halo from the source itself); the rows are gathered
    to rank 0 into out_rows ([n_frames, n], rank 0 only; None elsewhere).

    make_rows(f_lo, f_hi) -> tensor [f_hi - f_lo, n] on `device`, the spectra of frames f_lo..f_hi-1."""
    dtype = torch.float32 if dtype is None else dtype

    def consume(a, b, rows):
        if rows.data_ptr() != out_rows[a:b].data_ptr():     # rank 0 may produce its own rows in place
            out_rows[a:b].copy_(rows)

    return gather_chunked(n_frames, (n,), dtype, make_rows, consume, dist=dist, torch=torch, device=device,
                          n_chunks=n_chunks)


def checksum(torch, t):
    """Position-sensitive 64-bit checksum of a tensor's bytes (sum of (2 i + 1) * word_i over its little-endian 64-bit
    words, modulo 2^64; a tail shorter than a word is zero-padded), computed where the tensor lives.  bench.py's
    multi-GPU leg compares the checksum every member takes of its own rows before they travel with the one rank 0
    takes of what arrived, so that a chunk placed at the wrong offset or a byte lost on the wire shows up in the
    driver's record."""
    b = t.contiguous().view(torch.uint8).reshape(-1)
    pad = (-b.numel()) % 8
    if pad:
        b = torch.cat([b, torch.zeros(pad, dtype=torch.uint8, device=b.device)])
    w = b.view(torch.int64)
    total = 0
    piece = 1 << 22
    for s in range(0, w.numel(), piece):
        e = min(w.numel(), s + piece)
        idx = torch.arange(s, e, dtype=torch.int64, device=w.device)
        total += int((w[s:e] * (2 * idx + 1)).sum())          # int64 arithmetic wraps on the CPU and on the GPU alike
    return total & 0xFFFFFFFFFFFFFFFF


def tiles_of_image(image, a, b, n):
    """Tiles a..b-1 of a stitched image of side-by-side tiles (WIDTH_STEP = N) as the [b - a, H, N] stack their owner holds."""
    h = image.shape[0]
    return image[:, a * n: b * n].reshape(h, b - a, n).permute(1, 0, 2).contiguous()


def numpy_tiles_to_torch(torch, tiles, device):
    return torch.from_numpy(np.ascontiguousarray(tiles)).to(device)
