#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X IQ-FFT spectrum path on BASELINE.json's headline config.

  python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (u8 IQ -> flip -> (-1)^n -> FFT -> |X| + DC patch) over one
batch of synthetic int8 IQ that is already resident in HBM: by default BASELINE.json configs[2],
"Batched 8192-pt FFT, 4096 frames synthetic uint8 IQ, 1 MI355X (HBM-bound roofline run)".
Successive steps rotate through several independent input/output buffer sets (each 192 MiB) so
that no step can be served from the 256 MiB Infinity Cache.  With N > 1 (one process per GPU,
launched by torch.distributed.run) every rank transforms its own batch -- whole frames shard
across GPUs with no data-path collective -- and `value` is the aggregate over all ranks (weak
scaling).  Rank 0 prints ONE JSON line.

How the timed region is taken: an untimed clock-settle loop (64-launch blocks until three consecutive block times agree within
1 %, at most --prewarm seconds; config.clock_prewarm_s), the W warm-up steps, barrier + synchronise, then EXACTLY K timed steps,
synchronise + barrier: `value` / `ms_per_step` / `roofline` are that FIRST region.  --regions (9) identical regions follow it and
are reported beside it (headline_regions_ms, official_over_median): how representative the one graded sample was.

Extra objects on that line:
  roofline      dominant kernel vs the HBM roofline: algorithmic bytes per launch
                (2*hop + 4*N per frame) / average launch duration measured with HIP events on
                the launch stream over the timed region.
  cpu_baseline  the reference-shaped CPU loop (oracle/fsea_oracle.c, kind "port") timed on this
                host's cores on a bounded sample of the same workload; the transform inside it is
                an FFTW3-API library's when the host has one (libfftw3 is not installed in this
                image, Intel MKL's FFTW3 interface is), else the oracle's own.  A reported
                baseline, not the target.
  extra         the same measurement at N=1024 (the other size BASELINE.json names), and short runs of
                BASELINE.json's other single-GPU configurations (16384-point 50 %-overlap STFT, the
                fft-batch-broad sweep with stitch), the PCIe-inclusive host-buffer entry point, and BASELINE
                config 2 (nrf_fft(1024, 1024): nrf_fft_process + nrf_fft_get_buffer per rendered frame through
                libfsea_nrf.so; the oracle's restatement of the reference's CPU loop is cpu_baseline.nrf_stream).

PyTorch is plumbing only here: device buffers, streams, torch.distributed.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (fft_size, frames per batch, hop)
    "batch8192x4096": (8192, 4096, 8192),
    "batch1024x32768": (1024, 32768, 1024),
    "batch4096x8192": (4096, 8192, 4096),
    "stft16384x8191": (16384, 8191, 8192),
}


EXTRA_REGIONS = 5   # timed regions per informational figure (median reported)


def numpy_rows(raw_u8, n_frames, n, hop, mode="mag", window=None):
    """Sanity guard for what was just timed, numpy only (the parity tests proper live in tests/; the
    oracle is used by bench.py in the cpu_baseline leg alone): raw HackRF bytes -> offset binary ->
    x[k] = (-1)^k u8/256 -> forward FFT -> magnitude with bin N/2 := bin N/2-1 (src/nrf.c:599-630),
    or the *5 dB pixels with the same patch (c/fft-batch-broad.c:106-121)."""
    out = []
    sign = 1.0 - 2.0 * (np.arange(n) & 1)
    if window:                                       # periodic cosine-sum taper beside the (-1)^n, rounded to f32 as the kernel's
        coef = {"hann": (0.5, 0.5), "hamming": (0.54, 0.46), "blackman": (0.42, 0.5, 0.08)}[window]
        w = sum((-1) ** k * a * np.cos(2 * np.pi * k * np.arange(n) / n) for k, a in enumerate(coef))
        sign = sign * w.astype(np.float32).astype(np.float64)
    for f in range(n_frames):
        u = (raw_u8[2 * f * hop: 2 * (f * hop + n)] ^ np.uint8(0x80)).astype(np.float64) / 256.0
        spec = np.fft.fft((u[0::2] + 1j * u[1::2]) * sign)
        if mode == "mag":
            row = np.abs(spec)
        else:
            row = np.clip(np.trunc(10.0 * np.log10(spec.real ** 2 + spec.imag ** 2 + 1e-20) * 5.0), 0, 255)
        row[n // 2] = row[n // 2 - 1]
        out.append(row)
    return np.stack(out)


def synth_batch(seed, n_bytes):
    """HackRF-style int8 IQ: Gaussian sigma=20 + complex tone at +fs/8, amp 40."""
    rng = np.random.default_rng(seed)
    n = n_bytes // 2
    out = np.empty(2 * n, dtype=np.int8)
    chunk = 1 << 22
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        t = np.arange(s, e, dtype=np.float64)
        ph = 2 * np.pi * 0.125 * t
        i = rng.normal(0, 20, e - s) + 40 * np.cos(ph)
        q = rng.normal(0, 20, e - s) + 40 * np.sin(ph)
        out[2 * s:2 * e:2] = np.clip(np.rint(i), -128, 127)
        out[2 * s + 1:2 * e:2] = np.clip(np.rint(q), -128, 127)
    return out.view(np.uint8)


def settle_clock(torch, step, limit_s, block=64, agree=0.01, need=3, floor_s=0.25):
    """Untimed steps in blocks of `block` launches until `need` consecutive block times agree within `agree` (max over
    min of the last `need`) and at least min(floor_s, limit_s) seconds have passed (rounds 3-5 ran a fixed 0.25 s: never
    less than that), for at most `limit_s` seconds.  Returns (seconds spent, blocks run); (0.0, 0) if limit_s <= 0."""
    if limit_s <= 0:
        return 0.0, 0
    t_pre = time.perf_counter()
    torch.cuda.synchronize()
    times, k = [], 0
    while True:
        t0 = time.perf_counter()
        for _ in range(block):
            step(k)
            k += 1
        torch.cuda.synchronize()
        now = time.perf_counter()
        times.append(now - t0)
        last = times[-need:]
        if (len(times) > need and max(last) <= (1.0 + agree) * min(last)       # (the first block is never one of the three)
                and now - t_pre >= min(floor_s, limit_s)):
            break
        if now - t_pre >= limit_s:
            break
    return time.perf_counter() - t_pre, len(times)


def run_gpu(args, workload, rank, world, dist, torch, steps, warmup, sets, repeats=1, window=None, official_first=False):
    """K = `steps` timed steps of `workload` behind `warmup` untimed ones.  repeats > 1: the timed region is measured that
    many times back to back.  official_first (the headline): the FIRST region is the reported one -- the contract's W warm-ups
    + exactly K timed steps -- and the later, identical regions are reported beside it (`regions`); otherwise (the
    informational `extra` runs) the MEDIAN region is reported, so that a 20-step region of a side workload is not one
    sample.  window: a taper name for fsea_plan_set_window, or None."""
    from frequensea_amd import fsea

    n, frames, hop = WORKLOADS[workload]
    dev = torch.device("cuda", torch.cuda.current_device())
    variant = os.environ.get("FSEA_BENCH_VARIANT")               # tuning hook; unset = the product kernel
    if variant:
        fsea.use_tune_library()                                  # variants live in libfsea_hip_tune.so only
    plan = fsea.Plan(n, hop=hop, mode=fsea.MODE_MAG_F32, device=dev.index, variant=variant)
    if window:
        plan.set_window(window)
    in_bytes = plan.in_bytes(frames)
    host = synth_batch(3 + 1000 * rank, in_bytes)
    ins, outs = [], []
    for s in range(sets):
        t_in = torch.from_numpy(np.roll(host, 2 * 8 * s)).to(dev)   # distinct contents per set
        ins.append(t_in)
        outs.append(torch.empty(frames * n, dtype=torch.float32, device=dev))
    stream = torch.cuda.current_stream().cuda_stream

    def step(k):
        s = k % sets
        plan.exec_device(ins[s].data_ptr(), frames, outs[s].data_ptr(), flip=True, stream=stream)

    # clock settle loop (untimed, before the official warm-up; the seconds it took are named in config.clock_prewarm_s):
    # 64-launch blocks until three consecutive block times agree within 1 %, bounded by --prewarm seconds.  The shader
    # clock needs tens to hundreds of milliseconds of load to reach the package-power-capped state every later region
    # of this process sees; a fixed 0.25 s (rounds 3-5) left the official K steps on the governor's transient on some
    # boxes (BENCH_r05: the official 20-step region 7 % above its own process's medians).  Settling LOWERS the number
    # (the settled clock is the capped one); --prewarm 0 switches it off.
    prewarm_s, settle_blocks = settle_clock(torch, step, args.prewarm)
    # the events exist (and their record / elapsed_time paths have run once) BEFORE the clock starts: creating the first
    # timing events of the process inside the official region cost it 20-50 us of host time that no later region paid
    # (official_over_median 1.02-1.05 by wall against 1.00-1.01 by events, profiles/r06_official_over_median.txt)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(1, repeats))]
    for e0, e1 in events:
        e0.record()
        e1.record()
    torch.cuda.synchronize()
    events[0][0].elapsed_time(events[0][1])
    for k in range(warmup):
        step(k)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    walls, kernels = [], []
    for rep in range(max(1, repeats)):
        ev0, ev1 = events[rep]
        t0 = time.perf_counter()
        ev0.record()
        for k in range(steps):
            step(warmup + rep * steps + k)
        ev1.record()
        torch.cuda.synchronize()
        # this rank's own K steps are done: its clock stops here; the closing barrier brackets the region, the
        # job's time is the MAX over ranks (below), and a collective's own latency is not charged to the steps
        t1 = time.perf_counter()
        walls.append(t1 - t0)
        kernels.append(ev0.elapsed_time(ev1) / steps)  # events on the launch stream
        if rep == 0 and official_first and dist is not None:
            dist.barrier()                             # the contract's closing bracket of THE timed region
            torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    if official_first:
        # the headline: `value` is the FIRST K-step region (the bench contract: W warm-ups, then exactly K timed steps);
        # the identical regions behind it only say how representative that one sample was
        wall, kernel_ms = float(walls[0]), float(kernels[0])
    else:
        wall = float(np.median(walls))
        kernel_ms = float(np.median(kernels))
    regions = {"wall_ms_per_step": [1e3 * w / steps for w in walls], "events_ms_per_step": [float(x) for x in kernels]}
    if dist is not None:
        tt = torch.tensor([wall, kernel_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, kernel_ms = float(tt[0]), float(tt[1])
    # Informational only (never `value`): the same steps issued alternately on two streams, so that
    # one launch's drain overlaps the next one's ramp -- what a double-buffered consumer would see.
    two_stream = None
    if world == 1 and (args.two_stream or not args.no_extra) and workload.startswith("batch"):
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        torch.cuda.synchronize()
        # the same clock pre-warm as the one-stream figure gets, on the two streams: 20 launches from a chip that has
        # idled through the copies above run at a boost clock the sustained loop never sees (102 M against 91 M frames/s
        # at 8192 points; scripts/two_stream_lengths.py, profiles/r04_two_stream_lengths.txt)
        t_pre = time.perf_counter()
        k = 0
        while k < max(8, warmup) or time.perf_counter() - t_pre < min(args.prewarm, 0.25):
            for _ in range(64):
                s = k % sets
                plan.exec_device(ins[s].data_ptr(), frames, outs[s].data_ptr(), flip=True, stream=streams[k % 2].cuda_stream)
                k += 1
            torch.cuda.synchronize()
        rates = []
        for rep in range(max(3, repeats)):
            t2 = time.perf_counter()
            for k in range(steps):
                s = k % sets
                plan.exec_device(ins[s].data_ptr(), frames, outs[s].data_ptr(), flip=True, stream=streams[k % 2].cuda_stream)
            torch.cuda.synchronize()
            rates.append(frames * steps / (time.perf_counter() - t2))
        two_stream = float(np.median(rates))
        plan.exec_device(ins[0].data_ptr(), frames, outs[0].data_ptr(), flip=True, stream=stream)
        torch.cuda.synchronize()
    sample = outs[0][: 4 * n].cpu().numpy().reshape(4, n)
    kname = plan.kernel_name
    grid = plan.grid(frames)
    plan.close()
    return dict(n=n, frames=frames, hop=hop, wall=wall, kernel_ms=kernel_ms, kernel=kname, grid=grid, two_stream=two_stream,
                sample=sample, host_head=np.roll(host, 0)[: 2 * 4 * hop + 2 * n], steps=steps, regions=regions,
                prewarm_s=prewarm_s, settle_blocks=settle_blocks)


def pinned_checksums():
    """tests/golden/bench_job_checksums.json: the checksums of the config-4 stitched image and of the config-5 stream's rows
    as THIS code makes them on one GPU, committed after tests/test_gpu_bench_jobs.py compared exactly that image / those
    rows with the oracle, every row.  A gathered checksum that equals the pinned one therefore says the
    bytes on rank 0 are the oracle-verified ones, not merely what the members sent.  {} when the file is absent."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "bench_job_checksums.json")) as fp:
            return json.load(fp)
    except (OSError, ValueError):
        return {}


def run_broad(args, rank, world, dist, torch, steps, warmup, repeats=1, keep=None):
    """the fft-batch-broad sweep -- 512 centre frequencies x 256 frames x 4096-pt,
    u8 dB tiles (DB5 + DC fix), centre frequencies sharded over the ranks, tiles gathered to rank 0 over
    RCCL chunk by chunk (overlapped with the next chunk's FFT) and max-composited into the stitched image
    (c/fft-stitch-broad.c).  Strong scaling: the sweep is fixed, a step is the whole sweep including gather
    and stitch.  Two regimes, named in config.regime:
      resident  the captures are in HBM when the step starts (gather-to-one-root is then link-bound);
      ingest    the captures start in pinned host memory and every rank copies its shard over its own PCIe
                link inside the step, chunk by chunk on a copy stream ahead of the FFT."""
    from frequensea_amd import fsea, sweep

    n, rows, tiles = 4096, 256, 512
    dev = torch.device("cuda", torch.cuda.current_device())
    lo, hi = sweep.partition(tiles, world, rank)
    plan = fsea.Plan(n, hop=n, mode=fsea.MODE_DB5_U8_DCFIX, device=dev.index)
    gen = torch.Generator(device=dev)
    tile_bytes = 2 * rows * n
    samples = (hi - lo) * rows * n
    iq = torch.empty(2 * samples, dtype=torch.int8, device=dev)
    for f in range(lo, hi):                                     # int8 Gaussian sigma=20, generated on device, one seed per
        gen.manual_seed(4000000 + f)                            # centre frequency (seed = 4e6 + f): the sweep's
        s0 = (f - lo) * tile_bytes                              # captures, and with them the stitched image, are the same
        iq[s0:s0 + tile_bytes] = torch.clamp(torch.round(torch.randn(tile_bytes, generator=gen, device=dev) * 20.0),
                                             -128, 127).to(torch.int8)   # however many ranks share the work
    ingest = args.regime == "ingest"
    host_iq = None
    if ingest:
        host_iq = torch.empty(2 * samples, dtype=torch.int8, pin_memory=True)
        host_iq.copy_(iq)
        iq.zero_()                                              # every step has to bring the bytes in again
    px = torch.empty((hi - lo, rows, n), dtype=torch.uint8, device=dev)
    compute = torch.cuda.current_stream()
    copy_stream = torch.cuda.Stream() if ingest else None
    # chunks exist to overlap the gather (and the H2D ingest) with the FFT; one GPU with resident input has neither
    n_chunks = args.chunks if (world > 1 or ingest) else 1
    # consecutive chunks are transformed ALTERNATELY ON TWO STREAMS (include/fsea.h: fsea_stream_create): chunk j + 1 ramps up
    # under chunk j's drain; the stream that gathers and stitches (`compute`) waits for each chunk where it consumes it
    lanes = [torch.cuda.Stream(), torch.cuda.Stream()] if n_chunks > 1 else None
    issued = [0]
    step_no = [0]
    lane_step = [-1, -1]

    def launch_stream():
        if lanes is None:
            return compute
        k = issued[0] % 2
        s = lanes[k]
        if lane_step[k] != step_no[0]:
            # each lane's FIRST launch of a step waits for `compute`: this step's image (allocated, possibly zero-filled, on
            # `compute` inside run_sweep before the first chunk is produced) and the previous step's sends of the buffer the
            # launch is about to overwrite (only the first chunk of a step used to wait)
            s.wait_stream(compute)
            lane_step[k] = step_no[0]
        issued[0] += 1
        return s

    def make_tiles(a, b):                                       # tiles a..b-1 of the sweep, one launch
        s0, s1 = (a - lo) * tile_bytes, (b - lo) * tile_bytes
        st = launch_stream()
        if ingest:
            with torch.cuda.stream(copy_stream):
                iq[s0:s1].copy_(host_iq[s0:s1], non_blocking=True)
                ready = torch.cuda.Event()
                ready.record()
            st.wait_event(ready)
        plan.exec_device(iq.data_ptr() + s0, (b - a) * rows, px.data_ptr() + (a - lo) * rows * n, flip=True, stream=st.cuda_stream)
        if st is not compute:
            compute.wait_stream(st)                             # the consumer (send / composite) runs on `compute`
        return px[a - lo: b - lo]

    def write_tiles(image, a, b):                               # rank 0's own tiles, straight into the stitched image
        s0 = (a - lo) * tile_bytes
        st = launch_stream()
        if ingest:
            with torch.cuda.stream(copy_stream):
                iq[s0:(b - lo) * tile_bytes].copy_(host_iq[s0:(b - lo) * tile_bytes], non_blocking=True)
                ready = torch.cuda.Event()
                ready.record()
            st.wait_event(ready)
        plan.exec_tiled_device(iq.data_ptr() + s0, (b - a) * rows, image.data_ptr(), image.shape[0], image.shape[1], a * n,
                               rows, n, flip=True, stream=st.cuda_stream)
        if st is not compute:
            compute.wait_stream(st)

    stream = compute.cuda_stream

    def composite(image, tile, x):
        fsea.composite_max_device(image.data_ptr(), tile.data_ptr(), x, 0, n, rows, image.shape[1], image.shape[0], n,
                                  device=dev.index, stream=stream)

    def composite_stack(image, stack, count, first_x):
        fsea.stitch_tiles_device(image.data_ptr(), stack.data_ptr(), count, first_x, n, n, rows, image.shape[1],
                                 device=dev.index, stream=stream)

    def step():
        step_no[0] += 1
        return sweep.run_sweep(tiles, (rows, n), make_tiles, composite, dist=dist, torch=torch, device=dev,
                               composite_stack=composite_stack, n_chunks=n_chunks,
                               write_tiles=None if args.no_fused_stitch else write_tiles)

    for _ in range(max(warmup, 1)):
        img = step()
    torch.cuda.synchronize()
    # clock pre-warm, as for the headline workload (untimed; config.clock_prewarm_s)
    t_pre = time.perf_counter()
    while world == 1 and time.perf_counter() - t_pre < min(args.prewarm, 0.25):
        for _ in range(8):
            img = step()
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    walls, issues = [], []
    for rep in range(max(1, repeats)):
        t0 = time.perf_counter()
        for _ in range(steps):
            img = step()
        t_issued = time.perf_counter()    # the host has handed over all K steps (they run asynchronously behind it)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
        issues.append(t_issued - t0)
    wall = float(np.median(walls))        # own completion (rank 0: everything gathered); MAX over ranks below
    host_issue_ms = 1e3 * float(np.median(issues)) / steps
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    # FFT kernel alone on this rank's shard (HIP events on the launch stream), in the step's own form: rank 0's tiles
    # written into the stitched image
    if ingest:
        iq.copy_(host_iq)
    def kernel_only(count):
        for _ in range(count):
            if rank == 0 and not args.no_fused_stitch:
                plan.exec_tiled_device(iq.data_ptr(), (hi - lo) * rows, img.data_ptr(), img.shape[0], img.shape[1], lo * n,
                                       rows, n, flip=True, stream=stream)
            else:
                plan.exec_device(iq.data_ptr(), (hi - lo) * rows, px.data_ptr(), flip=True, stream=stream)
    kernel_only(20)                                             # (the chip idled while the host checked the image)
    torch.cuda.synchronize()
    kms = []
    for rep in range(max(3, repeats)):
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record()
        kernel_only(steps)
        k1.record()
        torch.cuda.synchronize()
        kms.append(k0.elapsed_time(k1) / steps)
    kernel_ms = float(np.median(kms))
    # one GPU, resident captures: consecutive sweeps issued alternately on two streams, each with its own image -- one
    # launch's drain under the next one's ramp, what a double-buffered consumer of independent sweeps gets
    two_stream_ms = None
    if world == 1 and not ingest and not args.no_fused_stitch and not args.no_extra:   # (--no-extra: profiling runs want the one-stream launches alone)
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        images = [img, torch.empty_like(img)]
        torch.cuda.synchronize()

        def sweep_on(k):
            plan.exec_tiled_device(iq.data_ptr(), tiles * rows, images[k % 2].data_ptr(), img.shape[0], img.shape[1], 0, rows, n,
                                   flip=True, stream=streams[k % 2].cuda_stream)
        for k in range(8):
            sweep_on(k)
        torch.cuda.synchronize()
        ts = []
        for rep in range(max(3, repeats)):
            t2 = time.perf_counter()
            for k in range(steps):
                sweep_on(k)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t2) / steps)
        two_stream_ms = 1e3 * float(np.median(ts))
        if not torch.equal(images[0], images[1]):
            raise SystemExit("bench broad: the two streams' stitched images differ")
        del images
    if dist is not None:
        tt = torch.tensor([wall, kernel_ms], dtype=torch.float64,
                          device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, kernel_ms = float(tt[0]), float(tt[1])
    # What arrived against what was sent: every rank recomputes its own tiles in ONE plain launch (no chunks, no tiled
    # stores) and takes their checksum; rank 0 takes the checksum of the same tiles as they sit in the stitched image after
    # the chunked gather of the last timed step.  A chunk at the wrong offset, a byte lost on the wire or a lane racing its
    # consumer shows up here, in the driver's own record.
    plan.exec_device(iq.data_ptr(), (hi - lo) * rows, px.data_ptr(), flip=True, stream=stream)
    torch.cuda.synchronize()
    sums = gather_objects(dist, world, (lo, hi, sweep.checksum(torch, px)))
    gathered_ok, gathered_sum = None, None
    if rank == 0:
        arrived = [sweep.checksum(torch, sweep.tiles_of_image(img, a, b, n)) for a, b, _ in sums]
        gathered_ok = all(x == y for x, (_, _, y) in zip(arrived, sums))
        gathered_sum = "%016x" % sweep.checksum(torch, img)     # of the whole stitched image: the same at every world size
        if not gathered_ok:
            bad = [r for r, (x, (_, _, y)) in enumerate(zip(arrived, sums)) if x != y]
            raise SystemExit("bench broad: the gathered tiles of rank(s) %s differ from what those ranks computed" % bad)
    # The gather by itself: the same chunked exchange with the tiles already computed (no FFT in the region), so that the
    # record holds what one peer -> root link carried per second.  Under gloo it is host staging + TCP, and says so.
    gather_only_ms = None
    if dist is not None:
        def produce_ready(a, b):
            return None if rank == 0 else px[a - lo: b - lo]

        def place(a, b, stack):
            img[:, a * n: b * n].view(rows, b - a, n).copy_(stack.permute(1, 0, 2))
        for it in range(2 + 5):
            if it == 2:
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                tg = time.perf_counter()
            sweep.gather_chunked(tiles, (rows, n), torch.uint8, produce_ready, place, dist=dist, torch=torch, device=dev,
                                 n_chunks=n_chunks)
        torch.cuda.synchronize()
        tt = torch.tensor([(time.perf_counter() - tg) / 5], dtype=torch.float64,
                          device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        gather_only_ms = 1e3 * float(tt[0])
    check = None
    if rank == 0:
        head = iq[: 2 * n * 2].cpu().numpy().view(np.uint8)
        want = numpy_rows(head, 2, n, n, mode="db5")
        got = img[:2, :n].cpu().numpy()
        check = int(np.abs(got.astype(np.int32) - want.astype(np.int32)).max())
        # the stitched image against the plain tile stack of this rank's shard (same kernel, untiled rows)
        plan.exec_device(iq.data_ptr(), (hi - lo) * rows, px.data_ptr(), flip=True, stream=stream)
        torch.cuda.synchronize()
        stack = px.view(hi - lo, rows, n)
        for k in (0, (hi - lo) // 2, hi - lo - 1):
            if not torch.equal(img[:, (lo + k) * n:(lo + k + 1) * n], stack[k]):
                raise SystemExit("bench broad: tile %d of the stitched image differs from the tile stack" % (lo + k))
        if check > 1:
            raise SystemExit("bench broad: stitched pixels differ from the numpy guard by %d" % check)
    frames_total = tiles * rows
    shard_frames = (hi - lo) * rows
    alg = (2 * n + n) * shard_frames
    line = {
        "metric": "fft_frames_per_sec_broad_sweep_n4096", "value": frames_total * steps / wall, "unit": "frames/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * wall / steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "broad: 512 centre freqs x 256 frames x 4096-pt, DB5_U8_DCFIX tiles, chunked gather "
                               "to rank 0 overlapped with the FFT; rank 0's own tiles are %s" %
                               ("stitched by the composite kernel" if args.no_fused_stitch else
                                "written in place by the FFT kernel (fsea_exec_u8_tiled_device), received ones copied in"),
                   "regime": ("ingest: captures start in pinned host memory, H2D inside the step (one PCIe link per GPU)"
                              if ingest else "resident: captures in HBM when the step starts (gather is xGMI-link-bound)"),
                   "gather_chunks": n_chunks, "clock_prewarm_s": min(args.prewarm, 0.25) if world == 1 else 0.0,
                   "parallelism": "centre frequencies sharded x%d, u8 tiles gathered by grouped send/recv" % world},
        "roofline": {"bound": "hbm", "achieved": alg / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                     "kernel": plan.kernel_name, "avg_launch_ms": kernel_ms, "algorithmic_bytes_per_launch": alg},
        "stitched_pixel_max_diff_vs_numpy_guard": check,
        # the step by its own clock and by the FFT kernel's: the difference is what the host spends per step that is not
        # hidden behind the device (first submission and last synchronise of the K-step region, spread over K)
        "ms_per_step_kernel_events": kernel_ms, "host_issue_ms_per_step": host_issue_ms,
        "roofline_frac_by_step_time": alg / (wall / steps) / 1e9 / HBM_PEAK_GBPS if world == 1 else None,
        "ms_per_step_two_streams": two_stream_ms, "timed_regions": max(1, repeats),
        "gathered_checksum_ok": gathered_ok, "gathered_checksum": gathered_sum,
    }
    pinned = pinned_checksums().get("broad_sweep_image")
    line["gathered_checksum_pinned"] = pinned
    line["gathered_checksum_matches_pinned"] = (gathered_sum == pinned) if (pinned and gathered_sum) else None
    if keep is not None:                                        # tests/test_gpu_bench_jobs.py: the job's own captures and image
        keep.update(image=img, iq=iq, tiles=tiles, rows=rows, n=n)
    if gather_only_ms is not None:
        peer_bytes = max((b - a) * rows * n for a, b in (sweep.partition(tiles, world, r) for r in range(1, world)))
        root_bytes = (tiles - (sweep.partition(tiles, world, 0)[1])) * rows * n
        line.update({"gather_only_ms": gather_only_ms, "gather_bytes_per_peer": peer_bytes,
                     "gather_gbps_per_link": peer_bytes / (gather_only_ms * 1e-3) / 1e9,
                     "gather_gbps_into_root": root_bytes / (gather_only_ms * 1e-3) / 1e9})
    plan.close()
    return line


def gather_objects(dist, world, obj):
    """[obj of rank 0, obj of rank 1, ...] on every rank (a small control-plane collective; [obj] without a job)."""
    if dist is None:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def stream_block(torch, dev, block, block_samples):
    """Block `block` of the synthetic 20 Msps stream of config C5 (seed 5): int8 Gaussian
    sigma = 20, generated on the device from a per-block seed, so that any rank can produce any part of
    the ONE global stream (its own frames plus the halo) without holding the rest."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(5000000 + block)
    x = torch.randn(2 * block_samples, generator=gen, device=dev) * 20.0
    return torch.clamp(torch.round(x), -128, 127).to(torch.int8)


def run_stft_stream(args, rank, world, dist, torch, steps, warmup, keep=None):
    """SURVEY 8(d)/(e) config C5 as specified: ONE synthetic stream, 16384-point frames at hop 8192 (50 %
    overlap); the frame range is cut into one contiguous piece per rank, every rank reads its samples plus
    the N - hop = 8192-sample halo it shares with its neighbour (nothing is exchanged), and the f32 rows are
    gathered to rank 0 chunk by chunk, overlapped with the next chunk's FFT.  Strong scaling: the stream is
    fixed, a step is the whole stream."""
    from frequensea_amd import fsea, sweep

    n, hop = 16384, 8192
    total_frames = args.stream_frames
    dev = torch.device("cuda", torch.cuda.current_device())
    f_lo, f_hi = sweep.partition(total_frames, world, rank)
    s_lo, s_hi = sweep.frame_sample_range(f_lo, f_hi, n, hop)
    block_samples = 1 << 20
    iq = torch.empty(2 * (s_hi - s_lo), dtype=torch.int8, device=dev)
    for blk in range(s_lo // block_samples, (s_hi + block_samples - 1) // block_samples):
        b0, b1 = blk * block_samples, (blk + 1) * block_samples
        a, b = max(b0, s_lo), min(b1, s_hi)
        iq[2 * (a - s_lo): 2 * (b - s_lo)] = stream_block(torch, dev, blk, block_samples)[2 * (a - b0): 2 * (b - b0)]
    plan = fsea.Plan(n, hop=hop, mode=fsea.MODE_MAG_F32, device=dev.index)
    if args.window:
        plan.set_window(args.window)
    out = torch.empty((total_frames, n), dtype=torch.float32, device=dev) if rank == 0 else None
    # rank 0 transforms its own frames straight into the gathered array; the others into a send buffer
    rows = out[f_lo:f_hi] if rank == 0 else torch.empty((f_hi - f_lo, n), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    n_chunks = args.chunks if world > 1 else 1
    compute = torch.cuda.current_stream()
    lanes = [torch.cuda.Stream(), torch.cuda.Stream()] if n_chunks > 1 else None   # consecutive chunks alternate (run_broad)
    issued = [0]

    step_no = [0]
    lane_step = [-1, -1]

    def make_rows(a, b):
        st = compute
        if lanes is not None:
            k = issued[0] % 2
            st = lanes[k]
            if lane_step[k] != step_no[0]:                      # first launch of a step on this lane: behind the previous
                st.wait_stream(compute)                         # step's sends of the rows it is about to overwrite
                lane_step[k] = step_no[0]
        issued[0] += 1
        plan.exec_device(iq.data_ptr() + 2 * (a * hop - s_lo), b - a, rows.data_ptr() + 4 * n * (a - f_lo), flip=True,
                         stream=st.cuda_stream)
        if st is not compute:
            compute.wait_stream(st)                             # the consumer (send / copy into place) runs on `compute`
        return rows[a - f_lo: b - f_lo]

    def step():
        step_no[0] += 1
        return sweep.run_stft(total_frames, n, make_rows, out, dist=dist, torch=torch, device=dev, n_chunks=n_chunks)

    for _ in range(max(warmup, 1)):
        step()
    torch.cuda.synchronize()
    # clock pre-warm, as for the sweep (untimed; one GPU only: with peers the steps carry collectives)
    t_pre = time.perf_counter()
    while world == 1 and time.perf_counter() - t_pre < min(getattr(args, "prewarm", 0.25), 0.25):
        for _ in range(4):
            step()
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0       # own completion (rank 0: everything gathered); MAX over ranks below
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    # the FFT kernel alone on this rank's shard, WARM: round 5 timed 5 launches straight after the barrier from a chip that
    # had idled through the host-side bookkeeping and reported 0.387 of the roofline for a kernel that runs at 0.45 once the
    # clock has settled (profiles/r06_stft_stream_shape.txt: cold 0.35-0.37, warm 0.45-0.47, however the stream is cut)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < min(getattr(args, "prewarm", 0.25), 0.25):
        plan.exec_device(iq.data_ptr(), f_hi - f_lo, rows.data_ptr(), flip=True, stream=stream)
        torch.cuda.synchronize()
    kms = []
    for _ in range(3):
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record()
        for _ in range(5):
            plan.exec_device(iq.data_ptr(), f_hi - f_lo, rows.data_ptr(), flip=True, stream=stream)
        k1.record()
        torch.cuda.synchronize()
        kms.append(k0.elapsed_time(k1) / 5)
    kernel_ms = float(np.median(kms))
    if dist is not None:
        tt = torch.tensor([wall, kernel_ms], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, kernel_ms = float(tt[0]), float(tt[1])
    # what arrived against what was sent (run_broad): every rank's rows, recomputed in one launch into a buffer of their own
    fresh = torch.empty((f_hi - f_lo, n), dtype=torch.float32, device=dev)
    plan.exec_device(iq.data_ptr(), f_hi - f_lo, fresh.data_ptr(), flip=True, stream=stream)
    torch.cuda.synchronize()
    sums = gather_objects(dist, world, (f_lo, f_hi, sweep.checksum(torch, fresh)))
    del fresh
    gathered_ok, gathered_sum = None, None
    if rank == 0:
        arrived = [sweep.checksum(torch, out[a:b]) for a, b, _ in sums]
        gathered_ok = all(x == y for x, (_, _, y) in zip(arrived, sums))
        gathered_sum = "%016x" % sweep.checksum(torch, out)     # of all rows of the stream: the same at every world size
        if not gathered_ok:
            bad = [r for r, (x, (_, _, y)) in enumerate(zip(arrived, sums)) if x != y]
            raise SystemExit("bench stft stream: the gathered rows of rank(s) %s differ from what those ranks computed" % bad)
    rel = None
    if rank == 0:                                               # rows 0, 1 and the last one against numpy
        head = stream_block(torch, dev, 0, block_samples)[: 2 * (hop + n)].cpu().numpy().view(np.uint8)
        want = numpy_rows(head, 2, n, hop, window=args.window)
        got = out[:2].cpu().numpy()
        rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
        if not rel <= 1e-6:
            raise SystemExit("bench stft stream: rows differ from the numpy guard (rel %.3e)" % rel)
    alg = (2 * hop + 4 * n) * (f_hi - f_lo)
    line = {
        "metric": "fft_frames_per_sec_stft_n16384_hop8192", "value": total_frames * steps / wall, "unit": "frames/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * wall / steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "msamples_per_sec": total_frames * steps / wall * hop / 1e6,
        "config": {"workload": "stft16384stream: one 20 Msps-style int8 stream, %d frames of 16384 points at hop 8192, %s, "
                               "f32 magnitude rows gathered to rank 0" %
                               (total_frames, "%s taper fused into pass 0" % args.window if args.window else "rectangular frames"),
                   "window": args.window or "rectangular (the reference)",
                   "regime": "resident: every rank's samples (frames + 8192-sample halo) are in HBM when the step starts",
                   "gather_chunks": n_chunks, "clock_prewarm_s": min(getattr(args, "prewarm", 0.25), 0.25) if world == 1 else 0.0,
                   "parallelism": "frame ranges x%d with an N - hop halo read redundantly, rows gathered by grouped send/recv" % world},
        "roofline": {"bound": "hbm", "achieved": alg / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                     "kernel": plan.kernel_name, "avg_launch_ms": kernel_ms, "algorithmic_bytes_per_launch": alg},
        "parity_rel_l2_first_rows": rel,
        "gathered_checksum_ok": gathered_ok, "gathered_checksum": gathered_sum,
    }
    # pinned for the stream as SURVEY 8(d) C5 specifies it (32767 frames, rectangular); any other length has no constant
    pinned = pinned_checksums().get("stft_stream_rows") if (total_frames == 32767 and not args.window) else None
    line["gathered_checksum_pinned"] = pinned
    line["gathered_checksum_matches_pinned"] = (gathered_sum == pinned) if (pinned and gathered_sum) else None
    if keep is not None:
        keep.update(rows=out, iq=iq, s_lo=s_lo, n=n, hop=hop, frames=total_frames)
    plan.close()
    return line


def host_path_rate(n, frames):
    """PCIe-inclusive rate of the same batch through fsea_exec_u8_host (pipelined copy-in / transform / copy-out, caller
    buffers that live across calls, pageable numpy memory); informational, never `value`."""
    from frequensea_amd import fsea
    plan = fsea.Plan(n)
    iq = synth_batch(3, 2 * n * frames)
    out = np.zeros((frames, n), np.float32)
    plan.exec_host_into(iq, frames, out)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        plan.exec_host_into(iq, frames, out)
        ts.append(time.perf_counter() - t0)
    plan.close()
    dt = float(np.median(ts))
    return {"host_path_frames_per_sec_n%d" % n: frames / dt, "host_path_ms_per_batch_n%d" % n: 1e3 * dt}


def nrf_stream_rate(n=1024, h=1024, frames=300):
    """BASELINE config 2 (configs[1]): the nrf_* API as lua/fft.lua:34-45 drives it -- nrf_fft_new(1024, 1024), then per
    rendered frame nrf_fft_process(one 262144-byte device block) + nrf_fft_get_buffer() (a fresh copy of the whole f64
    history) -- through libfsea_nrf.so, host-memory block in, malloc'd nut_buffer out.  Informational, never `value`."""
    from frequensea_amd import nrf
    L = nrf.nrf_lib()
    block = np.random.default_rng(2).integers(0, 256, nrf.NRF_BUFFER_SIZE_BYTES, dtype=np.uint8)
    buf = L.nut_buffer_new_u8(nrf.NRF_SAMPLES_LENGTH, 2, block.ctypes.data)
    fft = L.nrf_fft_new(n, h)
    for _ in range(20):
        L.nrf_fft_process(fft, buf)
    t0 = time.perf_counter()
    for _ in range(frames):
        L.nrf_fft_process(fft, buf)
    t1 = time.perf_counter()
    for _ in range(frames):
        L.nrf_fft_process(fft, buf)
        L.nut_buffer_free(L.nrf_fft_get_buffer(fft))
    t2 = time.perf_counter()
    L.nrf_fft_free(fft)
    L.nut_buffer_free(buf)
    return {"nrf_fft_%dx%d_process_us" % (n, h): (t1 - t0) / frames * 1e6,
            "nrf_fft_%dx%d_process_get_buffer_us" % (n, h): (t2 - t1) / frames * 1e6,
            "nrf_fft_%dx%d_rendered_frames_per_sec" % (n, h): frames / (t2 - t1)}


def cpu_nrf_stream(O, n=1024, h=1024, frames=40):
    """The same per-frame work as the reference does it on the CPU (src/nrf.c:598-635: unpack and centre ALL 131072
    samples of the block, one N-point transform, scroll the f64 history down one row, magnitudes, deep copy of the history),
    from the oracle's restatements (`O`, handed in by cpu_baseline: the one function of this file that imports it)."""
    block = np.random.default_rng(2).integers(0, 256, 262144, dtype=np.uint8)
    history = np.zeros((h, n))
    t0 = time.perf_counter()
    for _ in range(frames):
        x = O.unpack_center_u8(block)
        spec = O.fft_forward(np.ascontiguousarray(x[:n]))
        O.history_scroll(history, n, h)
        history[0, :] = O.mag_row(spec)
        out = history.copy()
    dt = time.perf_counter() - t0
    del out
    return {"process_get_buffer_us": dt / frames * 1e6, "rendered_frames_per_sec": frames / dt, "cores": 1,
            "what": "nrf_fft(%d,%d) per rendered frame, oracle restatement of src/nrf.c:598-635 (the reference is single-threaded "
                    "here; its powf(-1, ii) per sample is a sign select in the restatement, so the reference itself is slower)" % (n, h)}


def energy_per_frame(torch, n=8192, long_frames=16384, seconds=1.6, bench_shape_frames=4096):
    """Joules per frame of the headline kernel, rectangular and with a Hann taper: package power (rocm-smi, sampled while
    the kernel runs back to back on noise-like input) x HIP-event launch time / frames.  {} when rocm-smi cannot be read.
    The rectangular kernel is measured a second time in the bench's own launch shape (`bench_shape_frames` frames per
    launch, the K timed steps' form): a short launch spends part of its time ramping up and draining, so its average
    power sits below the cap that bounds the long launches' steady state -- the line says which figure belongs to which."""
    import re
    import subprocess
    import threading
    from frequensea_amd import fsea
    smi = "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(smi):
        return {}

    def power():
        try:
            out = subprocess.run([smi, "--showpower"], capture_output=True, text=True, timeout=10).stdout
            m = re.search(r"Package Power \(W\):\s*([0-9.]+)", out)
            return float(m.group(1)) if m else None
        except Exception:
            return None
    dev = torch.device("cuda", torch.cuda.current_device())
    host = synth_batch(9, 2 * n * long_frames)
    d_in = torch.from_numpy(host).to(dev)
    d_out = torch.empty(long_frames * n, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    res = {}
    for tag, window, frames in (("rect", None, long_frames), ("hann", "hann", long_frames), ("rect_bench_shape", None, bench_shape_frames)):
        plan = fsea.Plan(n, device=dev.index)
        if window:
            plan.set_window(window)
        watts, stop = [], []

        def sampler():
            time.sleep(0.5)                                # clocks and power settle
            while not stop:
                p = power()
                if p is not None:
                    watts.append(p)
                time.sleep(0.15)
        th = threading.Thread(target=sampler)
        th.start()
        ms = []
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pieces = max(1, long_frames // frames)       # short launches walk through the whole 768 MiB like the long one does
            per_region = 20 * pieces
            for k in range(per_region):
                off = (k % pieces) * frames
                plan.exec_device(d_in.data_ptr() + 2 * n * off, frames, d_out.data_ptr() + 4 * n * off, flip=True, stream=stream)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / per_region)
        stop.append(1)
        th.join()
        plan.close()
        if not watts:
            return {}
        w, t = float(np.median(watts)), float(np.median(ms[len(ms) // 3:]))
        res["energy_uj_per_frame_n8192_" + tag] = w * t * 1e-3 / frames * 1e6
        res["package_power_w_n8192_" + tag] = w
        res["launch_ms_n8192_" + tag] = t
    res["energy_note"] = ("rocm-smi package power x HIP-event launch time / frames, launches back to back for %.1f s: rect / hann "
                          "= %d-frame launches (steady state), rect_bench_shape = %d-frame launches (the timed steps' shape)"
                          % (seconds, long_frames, bench_shape_frames))
    return res


def skeleton_rates():
    import subprocess
    script = os.path.join(ROOT, "scripts", "skeleton_rates.py")
    lib = os.path.join(ROOT, "frequensea_amd", "libfsea_hip_tune.so")
    if not (os.path.exists(script) and os.path.exists(lib)):
        return None
    try:
        out = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=300)
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        return None


def effective_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (the GPU
    boxes show 256 logical CPUs under a quota of 16; more threads than that only get throttled)."""
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    note = ""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            text = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = text[0], float(text[1])
            else:
                quota, period = text[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                q = max(1, int(float(quota) / period))
                if q < cpus:
                    note = "cgroup CPU quota %d of %d visible CPUs" % (q, cpus)
                    cpus = q
            break
        except (OSError, ValueError, IndexError):
            continue
    return cpus, note


def _cpu_model():
    """Model name of the host's CPU (reported next to the core count), or None."""
    try:
        with open("/proc/cpuinfo") as fp:
            for line in fp:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline(n, hop, cores, budget_s, nrf_stream=False):
    """The reference-shaped CPU loop (flip -> unpack/centre -> FFT -> magnitude, oracle/fsea_oracle.c)
    timed on this host, bounded sample of about `budget_s` core-seconds.  The transform is done by an
    FFTW3-API library when the host has one -- the reference's own calls, fftw_plan_dft_1d +
    fftw_execute, one plan per thread -- i.e. FFTW itself, or Intel MKL through its FFTW3 interface;
    otherwise by the oracle's radix-2 f64 FFT.  kind "port" either way: the loop is this repo's."""
    from oracle import oracle as O

    fftw = O.find_fftw_api()
    buf_frames = max(2048, 16 * cores)                       # >= 16 frames per thread and pass
    iq = synth_batch(3, 2 * ((buf_frames - 1) * hop + n))
    if fftw is None:
        def run(frames_, threads, passes=1):
            return sum(O.time_mag_rows(iq, frames_, n, hop, threads=threads) for _ in range(passes))
        fft_name = "oracle/fsea_oracle.c radix-2 f64 FFT (no FFTW3-API library on this host)"
    else:
        def run(frames_, threads, passes=1):                 # threads and plans are made once per call
            return O.time_mag_rows_fftw(fftw, iq, frames_, n, hop, threads=threads, passes=passes)[0]
        fft_name = ("FFT by %s through the FFTW3 API (fftw_plan_dft_1d FFTW_MEASURE + fftw_execute, as src/nrf.c:562-615)"
                    % ("Intel MKL (%s)" % fftw if "mkl" in fftw else fftw))
    probe_frames = 64
    run(probe_frames, 1)                                     # library load, planner
    per_frame = max(run(probe_frames, 1) / probe_frames, 1e-7)
    f1 = min(buf_frames, max(64, int(2.0 / per_frame)))
    one_thread = f1 / run(f1, 1)
    reps = int(max(1, round(budget_s / per_frame / buf_frames)))
    frames = reps * buf_frames
    run(min(buf_frames, 16 * cores), cores)                  # warm the cores
    tm = run(buf_frames, cores, reps)
    out = dict(value=frames / tm, unit="frames/s", cores=cores, kind="port",
               sample="%d frames (%d passes over a %d-frame synthetic N=%d batch, frames sharded over %d threads, "
                      "one plan per thread); %s" % (frames, reps, buf_frames, n, cores, fft_name),
               one_thread=one_thread, host_cpus_visible=os.cpu_count(), cpu_model=_cpu_model())
    if fftw is not None:                                     # the oracle's own FFT beside it, short sample
        t = O.time_mag_rows(iq, buf_frames, n, hop, threads=cores)
        out["oracle_fft_value"] = buf_frames / t
    if nrf_stream:
        out["nrf_stream"] = cpu_nrf_stream(O)
    return out


def multi_gpu_leg(args, rank, world, dist, torch, stage, flat=None):
    """The two BASELINE workloads that have a real exchange step, in the form north_star scales them, run by the SAME
    command line the driver uses for its scaling record (`bench.py --gpus N`), after the headline's timed steps:

      config 4  the fft-batch-broad sweep (c/fft-batch-broad.c:176-206: the centre-frequency loop; c/fft-stitch-broad.c:62-87:
                the stitch) -- centre frequencies sharded over the ranks, u8 tiles gathered to rank 0 over RCCL chunk by
                chunk under the next chunk's FFT, stitched image on rank 0 -- in both regimes of SURVEY 8(e): captures
                resident in HBM, and captures ingested over every rank's own PCIe link inside the step;
      config 5  one 16384-point 50 %-overlap stream, frame ranges with a redundantly read halo, f32 rows gathered to rank 0.

    Both are strong-scaling figures (the job is fixed, a step is the whole job including gather and stitch) and live in
    `extra`; `value` stays the headline's.  `stage` (a one-element list) names what is running, for the watchdog's message;
    `flat` is filled in place stage by stage, so that a leg that hangs half-way still reports the stages it finished.
    Returns (flat keys for `extra`, {name: full line} for `extra.multi_gpu_lines`)."""
    import copy
    backend = dist.get_backend() if dist is not None else None
    flat = {} if flat is None else flat                      # filled in place: what is in it when the watchdog fires is printed
    flat.update({"gather_backend": backend if backend else "none (one rank, nothing to gather)", "gather_chunks": None})
    lines = {}
    stage[0] = "communicator census"
    # what the communicator itself reports: a SUM of ones over its ranks, on the device under nccl (an RCCL all-reduce)
    if dist is not None:
        one = torch.ones(1, dtype=torch.int64, device=torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else "cpu")
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        flat["rccl_world"] = int(one[0])
        flat["torch_world_size"] = dist.get_world_size()
    else:
        flat["rccl_world"] = 1
        flat["torch_world_size"] = 1
    if backend == "nccl":
        try:
            flat["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            flat["rccl_version"] = None
    props = torch.cuda.get_device_properties(torch.cuda.current_device())
    ident = str(getattr(props, "uuid", None) or getattr(props, "pci_bus_id", None) or torch.cuda.current_device())
    devs = gather_objects(dist, world, (rank, torch.cuda.current_device(), ident))
    flat["rank_devices"] = ["rank %d: device %d (%s)" % d for d in devs]
    flat["distinct_gpus"] = len(set(d[2] for d in devs))
    R = EXTRA_REGIONS if world == 1 else 3
    regimes = {}
    for regime in ("resident", "ingest"):
        stage[0] = "broad sweep, %s regime" % regime
        a2 = copy.copy(args)
        a2.regime = regime
        br = run_broad(a2, rank, world, dist, torch, 20 if world == 1 else 10, 3, repeats=R)
        lines["broad_" + regime] = br
        flat["broad_sweep_ms_" + regime] = br["ms_per_step"]
        flat["broad_sweep_frames_per_sec_" + regime] = br["value"]
        flat["broad_sweep_%s_gathered_checksum_ok" % regime] = br["gathered_checksum_ok"]
        flat["broad_sweep_%s_gathered_checksum" % regime] = br["gathered_checksum"]
        flat["broad_sweep_%s_gathered_checksum_matches_pinned" % regime] = br["gathered_checksum_matches_pinned"]
        regimes["broad_sweep_ms_" + regime] = br["config"]["regime"]
        flat["gather_chunks"] = br["config"]["gather_chunks"]
        if "gather_only_ms" in br and regime == "resident":
            flat.update({k: br[k] for k in ("gather_only_ms", "gather_bytes_per_peer", "gather_gbps_per_link", "gather_gbps_into_root")})
    if "gather_only_ms" not in flat:
        flat.update({"gather_only_ms": None, "gather_bytes_per_peer": 0, "gather_gbps_per_link": None, "gather_gbps_into_root": None})
    stage[0] = "stft stream"
    a2 = copy.copy(args)
    a2.window = None
    st = run_stft_stream(a2, rank, world, dist, torch, 5, 2)
    lines["stft_stream"] = st
    flat["stft_stream_ms"] = st["ms_per_step"]
    flat["stft_stream_frames_per_sec"] = st["value"]
    flat["stft_stream_gathered_checksum_ok"] = st["gathered_checksum_ok"]
    flat["stft_stream_gathered_checksum"] = st["gathered_checksum"]
    flat["stft_stream_gathered_checksum_matches_pinned"] = st["gathered_checksum_matches_pinned"]
    regimes["stft_stream_ms"] = st["config"]["regime"]
    flat["regime"] = regimes
    flat["multi_gpu_note"] = ("strong scaling: the sweep (512 x 256 x 4096-pt -> one stitched u8 image on rank 0) and the stream "
                              "(%d x 16384-pt rows on rank 0) are fixed jobs, a step is the whole job including gather and stitch; "
                              "compare *_ms across the driver's N = 1, 2, 4, 8 lines; *_gathered_checksum is of the whole stitched image / of all rows "
                              "on rank 0 and must be the same at every N (the captures are seeded per centre frequency / per stream block); "
                              "*_matches_pinned: it equals tests/golden/bench_job_checksums.json, the checksum of the image / rows "
                              "tests/test_gpu_bench_jobs.py compared with the oracle row by row (null: no constant for this stream length).  gather_gbps_per_link = the largest peer's "
                              "bytes / the gather alone (tiles precomputed, all peers sending at once)" % args.stream_frames)
    flat["multi_gpu_lines"] = {k: {"ms_per_step": v["ms_per_step"], "frames_per_sec": v["value"], "steps": v["steps"],
                                   "kernel_ms_largest_shard": v["roofline"]["avg_launch_ms"],
                                   "kernel_roofline_frac": v["roofline"]["frac"], "kernel": v["roofline"]["kernel"]}
                               for k, v in lines.items()}
    stage[0] = "done"
    return flat, lines


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="batch8192x4096", choices=sorted(WORKLOADS) + ["broad", "stft16384stream"])
    ap.add_argument("--regime", default="resident", choices=["resident", "ingest"],
                    help="broad sweep: captures resident in HBM, or H2D ingest inside the timed region")
    ap.add_argument("--chunks", type=int, default=8, help="chunks of the overlapped gather (broad, stft16384stream)")
    ap.add_argument("--stream-frames", type=int, default=32767,
                    help="stft16384stream: frames in the stream (32767 = 2^28 samples, SURVEY 8(d) C5)")
    ap.add_argument("--sets", type=int, default=6, help="independent buffer sets rotated per step")
    ap.add_argument("--no-fused-stitch", action="store_true",
                    help="broad: rank 0 stitches its own tiles with the composite kernel instead of writing them in place")
    ap.add_argument("--prewarm", type=float, default=2.0,
                    help="upper bound, seconds, of the untimed clock-settle loop in front of the --warmup steps (64-launch blocks "
                         "until three consecutive ones agree within 1 %%; the seconds it took: config.clock_prewarm_s); 0 = none")
    ap.add_argument("--regions", type=int, default=9,
                    help="identical K-step regions measured BEHIND the official one (headline_regions_ms, official_over_median)")
    ap.add_argument("--window", default=None, choices=["hann", "hamming", "blackman"],
                    help="taper fused into pass 0 (fsea_plan_set_window); default: none = the reference's rectangular frames")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--two-stream", action="store_true",
                    help="(default since round 3, kept for old command lines) the steps issued alternately on two streams, in `extra`")
    ap.add_argument("--cpu-budget", type=float, default=16.0, help="core-seconds for the CPU sample")
    ap.add_argument("--no-multi-gpu-leg", action="store_true",
                    help="skip the sharded sweep + stream (RCCL gather) that follow the headline steps in `extra`")
    ap.add_argument("--multi-gpu-timeout", type=float, default=240.0,
                    help="seconds the multi-GPU leg may take before the line is printed without it (extra.multi_gpu_error)")
    args = ap.parse_args()

    # Kernel arguments in device memory instead of host-coherent memory: the first s_load of a launch
    # otherwise crosses PCIe (about 0.5 us per 54 us launch here).  A HIP runtime knob, read when HIP
    # initialises, so it has to be set before torch touches the device; libfsea_hip.so sets the same
    # default for hosts that load it first.
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    # multi-process GPU work on these hosts needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails with the legacy mode)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one rank per GPU; FSEA_BENCH_BACKEND=gloo lets several ranks share one GPU to exercise the
    # N>1 control path on a single-GPU box (RCCL refuses two ranks on one device)
    backend = os.environ.get("FSEA_BENCH_BACKEND", "nccl")
    device_index = local % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist_mod.init_process_group(backend=backend)
        dist = dist_mod
    if args.gpus != world:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    if args.workload in ("broad", "stft16384stream"):
        fn = run_broad if args.workload == "broad" else run_stft_stream
        line = fn(args, rank, world, dist, torch, max(1, args.steps), max(1, args.warmup))
        if rank == 0:
            print(json.dumps(line))
        if dist is not None:
            dist.destroy_process_group()
        return

    res = run_gpu(args, args.workload, rank, world, dist, torch, args.steps, args.warmup, args.sets, window=args.window,
                  repeats=1 + max(0, args.regions), official_first=True)
    n, frames, hop = res["n"], res["frames"], res["hop"]
    # The timed region is the K steps between the opening and the closing barrier + device synchronise; `value` and
    # `ms_per_step` are the HOST WALL CLOCK of that region (MAX over ranks), as the bench contract defines them and as every
    # BENCH_r0x line before round 3 did (round 3 took them from the HIP events instead, which reads 3-5 % higher at K = 20:
    # first-launch submission and final-synchronise latency fall away).  The two HIP events recorded on the
    # launch stream around the same K steps give the device's own time for them: `value_events` / `ms_per_step_events`,
    # and -- being the kernel's average launch duration -- the roofline figures.
    value = world * frames * args.steps / res["wall"]
    value_events = world * frames / (res["kernel_ms"] * 1e-3)
    alg_bytes = (2 * hop + 4 * n) * frames                      # 2*hop read + 4*N written
    achieved = alg_bytes / (res["kernel_ms"] * 1e-3) / 1e9
    line = {
        "metric": "fft_frames_per_sec_n%d" % n,
        "value": value,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * res["wall"] / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "msamples_per_sec": value * hop / 1e6,
        "timing": "value / ms_per_step: host wall clock of the K timed steps between barrier + synchronise on both sides, MAX "
                  "over ranks; *_events: the two HIP events on the launch stream around the same K steps (device time)",
        "value_events": value_events,
        "ms_per_step_events": res["kernel_ms"],
        "config": {"workload": "%s: batched %d-pt FFT, %d frames per GPU per step, int8 IQ resident in HBM, "
                               "MAG_F32 epilogue (nrf_fft_process semantics)%s" %
                               (args.workload, n, frames, ", %s taper fused into pass 0" % args.window if args.window else ""),
                   "fft_size": n, "frames_per_step_per_gpu": frames, "hop": hop, "window": args.window or "rectangular (the reference)",
                   "buffer_sets": args.sets, "clock_prewarm_s": res["prewarm_s"], "clock_prewarm_limit_s": args.prewarm,
                   "clock_prewarm_blocks_of_64_launches": res["settle_blocks"],
                   "clock_prewarm_rule": "untimed 64-launch blocks until three consecutive block times agree within 1 % "
                                         "(at least 0.25 s, at most the limit), then the W warm-up steps",
                   "parallelism": "frames sharded x%d, no collective" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                     "kernel": res["kernel"], "avg_launch_ms": res["kernel_ms"],
                     "avg_launch_ms_source": "HIP events on the launch stream over the K timed steps",
                     "frac_by_step_time": alg_bytes / (res["wall"] / args.steps) / 1e9 / HBM_PEAK_GBPS,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "read_only_frac": (2 * hop * frames) / (res["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "grid_block_lds": list(res["grid"])},
    }
    # How representative the ONE official sample is: the same K steps measured `--regions` more times straight behind it
    # (rank 0's own clock; same buffers, same stream, nothing in between but the closing synchronise).  `value` /
    # `ms_per_step` / `roofline` stay the official (first) region; min / median / max are over the regions BEHIND it.
    rg = res["regions"]
    if len(rg["wall_ms_per_step"]) > 1:
        for key, src in (("headline_regions_ms", "wall_ms_per_step"), ("headline_regions_events_ms", "events_ms_per_step")):
            later = rg[src][1:]
            line[key] = {"official": rg[src][0], "min": float(np.min(later)), "median": float(np.median(later)),
                         "max": float(np.max(later)), "regions_behind_official": len(later)}
        line["official_over_median"] = line["headline_regions_ms"]["official"] / line["headline_regions_ms"]["median"]
        line["official_over_median_events"] = (line["headline_regions_events_ms"]["official"]
                                               / line["headline_regions_events_ms"]["median"])
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(traffic_file):
        try:
            tr = json.load(open(traffic_file))
            # a constant measured once per round with rocprofv3 PMC passes (scripts/pmc.sh), not by this
            # run; only quoted while kernel, grid and LDS footprint are the ones it was measured on
            if (tr.get("kernel") == res["kernel"] and tr.get("frames") == frames
                    and tr.get("grid_block_lds") == list(res["grid"])):
                line["roofline"]["traffic"] = tr.get("hbm_bytes_per_launch")
                line["roofline"]["traffic_source"] = tr.get("source")
                # The secondary ceilings SURVEY 8(d) names, from the same committed PMC passes (per launch, same guard):
                #   valu_issue_frac = vector instructions the launch issues / THIS run's launch time / the chip's measured issue
                #     rate for the kernel's dominant instruction at its occupancy (v_pk_fma_f32, two waves per SIMD:
                #     profiles/r01_valu_issue_rate_microbench.txt) -- how much of the time the vector pipes have work;
                #   lds_active_frac = cycles the CUs' LDS units were active / (CUs x the launch's shader cycles).
                if tr.get("sq_insts_valu_per_launch") and tr.get("valu_peak_wave_insts_per_s"):
                    line["roofline"]["valu_issue_frac"] = (tr["sq_insts_valu_per_launch"] / (res["kernel_ms"] * 1e-3)
                                                           / tr["valu_peak_wave_insts_per_s"])
                    line["roofline"]["valu_issue_note"] = tr.get("valu_issue_note")
                if tr.get("sq_lds_idx_active_per_launch") and tr.get("shader_cycles_per_launch") and tr.get("compute_units"):
                    line["roofline"]["lds_active_frac"] = (tr["sq_lds_idx_active_per_launch"]
                                                           / (tr["compute_units"] * tr["shader_cycles_per_launch"]))
                    line["roofline"]["lds_active_note"] = tr.get("lds_active_note")
        except Exception:
            pass

    if rank == 0:
        # correctness guard on what was just timed
        want = numpy_rows(res["host_head"], 4, n, hop, window=args.window)
        rel = float(np.linalg.norm(res["sample"] - want) / np.linalg.norm(want))
        line["parity_rel_l2_first_rows"] = rel
        if not rel <= 1e-6:
            raise SystemExit("bench: GPU rows differ from the numpy guard (rel %.3e)" % rel)

    br = None
    if args.workload == "batch8192x4096" and not args.window and not args.no_extra:
        start_guardian(line, rank)                                # from here on a hard failure still leaves the headline's line
    if world > 1 and args.workload == "batch8192x4096" and not args.window and not args.no_multi_gpu_leg and not args.no_extra:
        # The driver's N > 1 command line: after the headline's steps, the workloads north_star actually scales over a node
        mg_flat, mg_lines = guarded_multi_gpu_leg(args, rank, world, dist, torch, line)
        line["extra"] = mg_flat
    if world == 1 and not args.no_extra and args.workload == "batch8192x4096" and not args.window:
        # Everything in `extra` is informational.  Each figure: the same clock pre-warm as the headline, then the MEDIAN of
        # EXTRA_REGIONS timed regions of K' steps (HIP events per region) -- not one sample of a 20-step region.
        R = EXTRA_REGIONS
        ex_steps = max(20, min(args.steps, 200))
        ex = run_gpu(args, "batch1024x32768", rank, world, dist, torch, ex_steps, args.warmup, args.sets, repeats=R)
        exb = (2 * ex["hop"] + 4 * ex["n"]) * ex["frames"]
        line["extra"] = {"timed_regions_per_figure": R,
                         "fft_frames_per_sec_n1024": ex["frames"] / (ex["kernel_ms"] * 1e-3),
                         "msamples_per_sec_n1024": ex["frames"] / (ex["kernel_ms"] * 1e-3) * ex["hop"] / 1e6,
                         "roofline_frac_n1024": exb / (ex["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "kernel_n1024": ex["kernel"], "avg_launch_ms_n1024": ex["kernel_ms"]}
        # the same steps issued alternately on two streams (one launch's drain overlaps the next one's ramp): what a
        # double-buffered consumer of independent batches gets; informational, never `value` (host wall clock)
        line["extra"].update({"two_stream_frames_per_sec_n8192": res["two_stream"],
                              "two_stream_frames_per_sec_n1024": ex["two_stream"]})
        # the taper window north_star names, fused into pass 0 (fsea_plan_set_window): the headline batch and BASELINE config
        # 5's 50 %-overlap STFT with a Hann taper, each beside its rectangular twin measured the same way in this run
        # regions of at least 100 launches for this pair: a ratio of two 20-launch regions (1 ms each, from an idle chip) moved
        # between 0.94 and 0.99 on one build from box to box; 100 launches per region settle it (same launch shape)
        h_steps = max(100, min(args.steps, 200))
        r8 = run_gpu(args, "batch8192x4096", rank, world, dist, torch, h_steps, args.warmup, args.sets, repeats=R)
        h8 = run_gpu(args, "batch8192x4096", rank, world, dist, torch, h_steps, args.warmup, args.sets, repeats=R, window="hann")
        st_steps = max(50, min(args.steps, 100))
        st = run_gpu(args, "stft16384x8191", rank, world, dist, torch, st_steps, min(args.warmup, 20), 2, repeats=R)
        sh = run_gpu(args, "stft16384x8191", rank, world, dist, torch, st_steps, min(args.warmup, 20), 2, repeats=R, window="hann")
        for tag, run_ in (("hann_n8192", h8), ("stft16384_hann", sh)):
            want = numpy_rows(run_["host_head"], 4, run_["n"], run_["hop"], window="hann")
            relw = float(np.linalg.norm(run_["sample"] - want) / np.linalg.norm(want))
            if not relw <= 1e-6:
                raise SystemExit("bench: windowed rows (%s) differ from the numpy guard (rel %.3e)" % (tag, relw))
        stb = (2 * st["hop"] + 4 * st["n"]) * st["frames"]
        line["extra"].update({"stft16384_hop8192_frames_per_sec": st["frames"] / (st["kernel_ms"] * 1e-3),
                              "stft16384_roofline_frac": stb / (st["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                              "stft16384_kernel": st["kernel"],
                              "stft16384_hann_frames_per_sec": sh["frames"] / (sh["kernel_ms"] * 1e-3),
                              "stft16384_hann_roofline_frac": stb / (sh["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                              "stft16384_hann_kernel": sh["kernel"],
                              "stft16384_hann_over_rect": st["kernel_ms"] / sh["kernel_ms"],
                              "hann_n8192_frames_per_sec": h8["frames"] / (h8["kernel_ms"] * 1e-3),
                              "hann_n8192_roofline_frac": alg_bytes / (h8["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                              "hann_n8192_kernel": h8["kernel"],
                              "hann_n8192_over_rect": r8["kernel_ms"] / h8["kernel_ms"]})
        en = energy_per_frame(torch)
        line["extra"].update(en)
        if en:
            # which limit binds where, measured in both launch shapes: package power while the kernel runs back to back in
            # 16384-frame launches (steady state) and in the timed steps' own 4096-frame launches
            line["roofline"].update({"package_power_w_in_bench_shape": en.get("package_power_w_n8192_rect_bench_shape"),
                                     "package_power_w_long_launches": en.get("package_power_w_n8192_rect"),
                                     "package_power_cap_w": 1400.0,
                                     "frac_long_launches": (alg_bytes // frames) * 16384 / (en["launch_ms_n8192_rect"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
                                     if en.get("launch_ms_n8192_rect") else None,
                                     "energy_uj_per_frame_in_bench_shape": en.get("energy_uj_per_frame_n8192_rect_bench_shape"),
                                     "energy_uj_per_frame_long_launches": en.get("energy_uj_per_frame_n8192_rect"),
                                     "limit_note": "package power (rocm-smi) with the kernel running back to back: both launch shapes sit "
                                                   "within a few per cent of the 1400 W cap, so the shader clock is the governor's answer to "
                                                   "joules per frame in both; the 4096-frame launches of the timed steps pay more joules "
                                                   "per frame than the 16384-frame ones (ramp and tail burn power without finishing "
                                                   "frames): that difference is the gap between frac and frac_long_launches"})
        mg_flat, mg_lines = guarded_multi_gpu_leg(args, rank, world, dist, torch, line)
        line["extra"].update(mg_flat)
        br = mg_lines.get("broad_resident")
    if world == 1 and not args.no_extra and args.workload == "batch8192x4096" and not args.window and br is not None:
        line["extra"].update({"broad_sweep_1gpu_frames_per_sec": br["value"], "broad_sweep_1gpu_ms": br["ms_per_step"],
                              "broad_sweep_1gpu_kernel_ms": br["ms_per_step_kernel_events"],
                              "broad_sweep_1gpu_host_issue_ms": br["host_issue_ms_per_step"],
                              "broad_sweep_1gpu_ms_two_streams": br["ms_per_step_two_streams"],
                              "broad_sweep_roofline_frac": br["roofline"]["frac"],
                              "broad_sweep_roofline_frac_by_step_time": br["roofline_frac_by_step_time"],
                              "broad_sweep_note": "ms = host wall clock per whole-sweep step (K = 20, median of the timed regions); "
                                                  "kernel_ms = the tiled FFT launch by HIP events; the difference is the first "
                                                  "submission + last synchronise of a 20-step region (a few tens of us) spread "
                                                  "over its steps, plus Python's per-step issue time where that exceeds the kernel"})
    if world == 1 and not args.no_extra and args.workload == "batch8192x4096" and not args.window:
        line["extra"].update(host_path_rate(n, frames))
        line["extra"].update(nrf_stream_rate())
        # what bounds the headline kernel from above on THIS box, same launch shape and buffer rotation: its I/O skeleton
        # and a plain 1 : 2 read/write stream (tuning library, in a subprocess: scripts/skeleton_rates.py).  The subprocess
        # measures the product kernel the same way beside them: the RATIOS are what compares like with like.
        sk = skeleton_rates()
        if sk:
            line["roofline"].update({"io_skeleton_frac": sk["io_skeleton_frac"], "copy_frac": sk["copy_frac"],
                                     "io_skeleton_launch_ms": sk["io_skeleton_launch_ms"], "copy_launch_ms": sk["copy_launch_ms"],
                                     "product_frac_in_that_process": sk["product_frac"],
                                     "kernel_over_io_skeleton": sk["product_frac"] / sk["io_skeleton_frac"],
                                     "kernel_over_copy": sk["product_frac"] / sk["copy_frac"],
                                     "ceiling_note": "io_skeleton = this kernel's loads and row stores without LDS exchange and "
                                                     "butterflies (abl_io_nt); copy = a plain 16-bytes-in / 32-bytes-out nt "
                                                     "stream; all three measured alike in one subprocess (scripts/skeleton_rates.py: "
                                                     "0.25 s pre-warm, median of 5 rounds of 120 launches): compare the kernel with "
                                                     "them through kernel_over_*, not through `frac`"})

    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cores, quota_note = effective_cpus()
        cb = cpu_baseline(n, hop, cores, args.cpu_budget,
                          nrf_stream=not args.no_extra and args.workload == "batch8192x4096")
        if quota_note:
            cb["sample"] += "; " + quota_note
        line["cpu_baseline"] = cb
        line["gpu_over_cpu_all_cores"] = value / cb["value"]

    emit_line(line, rank)
    if dist is not None:
        if line.get("extra", {}).get("multi_gpu_error") or _LEG_FAILED[0]:
            # a rank left the leg by an exception: its peers may sit in a collective it never joined; the line is out, so
            # this rank leaves without the process group's closing handshake (which could wait for them)
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)
        dist.destroy_process_group()


_EMIT_LOCK = None
_EMITTED = [False]
_LEG_FAILED = [False]
_GUARDIAN = [None]

GUARDIAN_SCRIPT = r"""
import json, sys
provisional, final = None, None
for raw in sys.stdin:
    tag, _, body = raw.partition(" ")
    if tag == "PROVISIONAL":
        provisional = body
    elif tag == "FINAL":
        final = body
        break
if final is None and provisional is not None:
    try:
        d = json.loads(provisional)
        d.setdefault("extra", {})["multi_gpu_error"] = ("the bench process ended without printing its line (killed or aborted "
                                                        "inside the multi-GPU leg); this is the headline as measured before the leg")
        final = json.dumps(d) + "\n"
    except Exception:
        final = provisional
if final is not None:
    sys.stdout.write(final if final.endswith("\n") else final + "\n")
    sys.stdout.flush()
"""


def start_guardian(line, rank):
    """the headline is measured before the multi-GPU leg, and a HARD failure inside the leg (an RCCL watchdog's
    abort(), a GPU fault, SIGKILL / SIGTERM from the launcher after a peer died) cannot be caught by Python.  Rank 0 therefore
    hands the line as it stands to a small child process (no GPU, no torch: it only reads its stdin) which becomes THE printer:
    it prints the final line when main() delivers one, and the provisional one -- with extra.multi_gpu_error saying so -- if
    its stdin closes without.  One line either way."""
    import subprocess
    if rank != 0 or _GUARDIAN[0] is not None:
        return
    try:
        g = subprocess.Popen([sys.executable, "-c", GUARDIAN_SCRIPT], stdin=subprocess.PIPE, text=True)
        g.stdin.write("PROVISIONAL " + json.dumps(line) + "\n")
        g.stdin.flush()
        _GUARDIAN[0] = g
    except Exception as e:                                       # no guardian: main() prints by itself, as before
        print("bench.py: no guardian process (%s: %s)" % (type(e).__name__, e), file=sys.stderr)
        _GUARDIAN[0] = None


def emit_line(line, rank):
    """Rank 0 prints THE one JSON line, once -- whether main() got to its end or the multi-GPU leg's watchdog fired first
    (through the guardian process when there is one: a single printer, so that no failure order can print two lines)."""
    global _EMIT_LOCK
    import threading
    if _EMIT_LOCK is None:
        _EMIT_LOCK = threading.Lock()
    with _EMIT_LOCK:
        if _EMITTED[0]:
            return
        _EMITTED[0] = True
        if rank == 0:
            try:
                text = json.dumps(line)
            except Exception as e:                               # (a key of the wrong type from a half-finished stage)
                text = json.dumps({k: v for k, v in line.items() if k != "extra"})
                print("bench.py: line serialised without `extra` (%s: %s)" % (type(e).__name__, e), file=sys.stderr)
            g = _GUARDIAN[0]
            if g is not None:
                try:
                    g.stdin.write("FINAL " + text + "\n")
                    g.stdin.flush()
                    g.stdin.close()
                    if g.wait(timeout=20) == 0:
                        return
                except Exception:
                    pass                                         # the guardian is gone (or failed): print here
            print(text)
            sys.stdout.flush()


def guarded_multi_gpu_leg(args, rank, world, dist, torch, line):
    """multi_gpu_leg() under a guard: whatever happens in it -- an exception on this rank, a peer that died, a collective that
    never completes on a fabric this code has not met yet -- the headline's line (already measured; `value` is not touched
    by anything here) is printed, with the failure named in extra.multi_gpu_error.  A watchdog thread covers the case that
    cannot raise: after --multi-gpu-timeout seconds rank 0 prints the line as it stands and every rank leaves."""
    import threading
    import traceback
    stage = ["starting"]

    limit = [float(args.multi_gpu_timeout)]
    partial = {}

    done = threading.Event()

    def on_timeout():
        # runs on the timer's thread while the main thread may still be filling `partial`: every step is fenced, and the
        # process leaves whatever happens in here
        try:
            if done.is_set():                                    # the leg finished while the timer was firing
                return
            msg = "the multi-GPU leg did not finish within %.0f s (stage: %s); line printed without it" % (limit[0], stage[0])
            snap = None
            for _ in range(5):                                   # a dict that changes size under the copy raises: try again
                try:
                    snap = json.loads(json.dumps(dict(partial), default=str))
                    break
                except Exception:
                    time.sleep(0.01)
            line.setdefault("extra", {})
            if snap:
                line["extra"].update(snap)                       # the stages that did finish
            line["extra"]["multi_gpu_error"] = msg
            print("bench.py rank %d: %s" % (rank, msg), file=sys.stderr)
            emit_line(line, rank)
        finally:
            if not done.is_set():
                sys.stdout.flush()
                sys.stderr.flush()
                os._exit(0)
    dog = threading.Timer(limit[0], on_timeout)
    dog.daemon = True
    dog.start()
    flat, lines = partial, {}
    try:
        flat, lines = multi_gpu_leg(args, rank, world, dist, torch, stage, flat=partial)
        flat["multi_gpu_error"] = None
    except BaseException as e:                                   # SystemExit from the leg's own checks included
        if isinstance(e, KeyboardInterrupt):
            raise
        _LEG_FAILED[0] = True
        flat = dict(partial)                                     # the stages that did finish stay in the line
        flat["multi_gpu_error"] = "rank %d, stage %s: %s: %s" % (rank, stage[0], type(e).__name__, e)
        print("bench.py rank %d: multi-GPU leg failed in stage %s\n%s" % (rank, stage[0], traceback.format_exc()), file=sys.stderr)
    done.set()
    dog.cancel()
    if dist is not None and not _LEG_FAILED[0]:
        # did every rank get through?  (a rank that failed is not here: a short watchdog then ends the wait)
        limit[0], stage[0] = 45.0, "closing census (a peer left the leg early: see its stderr)"
        done.clear()
        dog = threading.Timer(limit[0], on_timeout)
        dog.daemon = True
        dog.start()
        try:
            oks = gather_objects(dist, world, flat.get("multi_gpu_error"))
            bad = [(r, m) for r, m in enumerate(oks) if m]
            if bad:
                flat["multi_gpu_error"] = "; ".join("rank %d: %s" % b for b in bad)
        except BaseException as e:
            _LEG_FAILED[0] = True
            flat["multi_gpu_error"] = "closing census: %s: %s" % (type(e).__name__, e)
    done.set()
    dog.cancel()
    return flat, lines


if __name__ == "__main__":
    main()
