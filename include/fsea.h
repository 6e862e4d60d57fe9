/*
 * fsea.h -- C ABI of libfsea_hip.so: the MI355X (gfx950) IQ-FFT spectrum path.
 *
 * This is the drop-in boundary.  It sits exactly where the reference calls
 * FFTW and runs its per-sample loops on the CPU (paths under /root/reference):
 *
 *   fsea_plan_create      replaces fftw_plan_dft_1d + buffer setup
 *                         src/nrf.c:562-564, c/fft-batch.c:140-145,
 *                         c/fft-batch-broad.c:167-172
 *   fsea_exec_u8_*        replaces byte flip + unpack/centre + fftw_execute +
 *                         magnitude / dB pixel loops
 *                         src/nrf.c:100-109 (flip), 599-614 (unpack, (-1)^n),
 *                         615 (fftw_execute), 619-630 (magnitude + DC patch);
 *                         c/fft-batch.c:62-69, 83-94; c/fft-batch-broad.c:64-71,
 *                         106-121
 *   fsea_exec_f64_host    the NUT_BUFFER_F64 input branch, src/nrf.c:607-609
 *   fsea_exec_u8_shifted_* nrf_freq_shifter_process in front of the FFT, fused into its load
 *                         src/nrf.c:843-866 (shifter) + 607-612 (F64 branch);
 *                         call pattern lua/fft-shifted.lua:52-55
 *   fsea_mean_magnitude_* the 100-row "interesting?" gate,
 *                         c/fft-batch-broad.c:81-98
 *   fsea_composite_max_*  tile compositing, c/fft-stitch.c:46-54,
 *                         c/fft-stitch-broad.c:28-36
 *   fsea_plan_set_window  a taper in the weight slot of the unpack loop: the reference weights each sample by
 *                         powf(-1, ii) alone (src/nrf.c:611-612, c/fft-batch.c:65-66), i.e. its window is
 *                         rectangular; this is the optional w[n] beside it, fused into the same conversion
 *   fsea_plan_destroy     replaces fftw_destroy_plan/fftw_free
 *                         src/nrf.c:637-642, c/fft-batch.c:147-152
 *
 * Plain C: pointers, sizes and ints only; no HIP or torch types.  Device
 * pointers and streams cross the boundary as void* (a hipStream_t, e.g.
 * torch.cuda.current_stream().cuda_stream).  Every function returns 0 on
 * success or a negative FSEA_E* code; fsea_last_error_string() describes the
 * last failure on the calling thread.  There is no CPU fallback: without a
 * usable gfx950 device plan creation fails with FSEA_ENODEVICE.
 */
#ifndef FSEA_H
#define FSEA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fsea_plan fsea_plan;
typedef struct fsea_history fsea_history;

/* Epilogue modes.  Output element type and row length are per mode. */
enum {
    /* sqrt(re^2+im^2) as f32, bin n/2 := bin n/2-1 (src/nrf.c:619-630). */
    FSEA_MODE_MAG_F32 = 0,
    /* clamp_u8(trunc(10*log10(re^2+im^2+1e-20)*10)) (c/fft-batch.c:83-94).  The f32 kernels form d = 100 log10(p) and
     * convert with v_cvt_pk_u8_f32 (round to nearest even) after lowering d by 0.5 - 2^-25: trunc(d) except for d within
     * ~8e-6 above an odd integer, where the pixel comes out one grey level low -- about one pixel in 2.5e5, inside the
     * stated pixel tolerance (exact on >= 99.9 %, +-1 elsewhere: the f32 logarithm itself moves more pixels than that
     * across an integer boundary); tests/test_emu_kernels.py::test_biased_pixel_rounding_equals_truncation_on_a_dense_sweep pins it.  The Bluestein /
     * four-step sizes truncate in a plain epilogue kernel. */
    FSEA_MODE_DB10_U8 = 1,
    /* same with *5 and pixel n/2 := pixel n/2-1 (c/fft-batch-broad.c:106-121). */
    FSEA_MODE_DB5_U8_DCFIX = 2,
    /* the complex spectrum itself, interleaved f32 re,im (what fft_out holds). */
    FSEA_MODE_COMPLEX_F32 = 3,
    /* magnitude without the DC patch. */
    FSEA_MODE_MAG_NODC_F32 = 4,
    /* 10*log10(re^2+im^2+1e-20) as f32 (log-magnitude, no quantisation). */
    FSEA_MODE_DB_F32 = 5
};

enum {
    FSEA_OK = 0,
    FSEA_EINVAL = -1,    /* bad argument (size, hop, alignment, mode) */
    FSEA_ENODEVICE = -2, /* no usable gfx950 device / HIP runtime failure at init */
    FSEA_ENOMEM = -3,
    FSEA_EHIP = -4       /* a HIP call failed; see fsea_last_error_string() */
};

/* Number of visible HIP devices (0 and FSEA_ENODEVICE when there are none). */
int fsea_device_count(int *count);

/* fft_size: powers of two from 32 to 16384 have kernels of their own (the scripts and tools of the reference use
 * 128 ... 16384).  fftw_plan_dft_1d (src/nrf.c:564) takes any size; the others run on those kernels:
 *   - powers of two from 32768 to 2^20: two passes of the kernels (four-step: n = n1 n2, column transforms, twiddle,
 *     row transforms) and three helper kernels;
 *   - everything else from 2 to 2^19: Bluestein's algorithm, two transforms of size 2^p >= 2 fft_size - 1 around a
 *     pointwise product with the chirp's spectrum.
 * Same modes, same tolerance.  Such a plan serves fsea_exec_u8_device, the three *_host entry points, the history ring
 * and the gate, but not the tiled and the frequency-shifted entry points (FSEA_EINVAL); it is a compatibility path, not
 * a tuned one: five to ten launches per batch through work buffers that belong to the plan, so launches of one such plan
 * on different streams run one after the other (ordered by events), and they cannot be captured into a graph.
 * Larger sizes fail with FSEA_EINVAL.
 * hop: samples between successive frame starts (hop == fft_size: back-to-back
 * frames as in c/fft-batch.c; hop < fft_size: overlapped STFT).  hop must be a
 * positive multiple of 8 for the sizes with kernels of their own, any positive number for the others.
 * device: HIP device ordinal. */
int fsea_plan_create(fsea_plan **plan, int fft_size, int hop, int mode, int device);
int fsea_plan_destroy(fsea_plan *plan);

/* Taper window.  With a window set, every u8 transform of the plan computes
 *   X[k] = sum_n (-1)^n w[n] (u8[n] / 256) e^{-2 pi i n k / fft_size}
 * -- the reference's unpack loop (src/nrf.c:601-614) with w[n] beside its powf(-1, ii); w == 1 is the reference
 * itself -- followed by the plan's epilogue unchanged (MAG and DB5 rows still copy bin n/2 - 1 into bin n/2: with a
 * taper the offset-binary DC term also reaches the neighbours of bin n/2, as it does in the reference's arithmetic).
 * The multiply is fused into the kernel's byte conversion (one packed multiply per sample; no extra pass, no extra
 * HBM traffic: the fft_size weights live in L2 / registers).
 *   w: fft_size floats, copied; NULL removes the window (rectangular, the un-windowed kernels).  Any finite values.
 * Applies to every transform of the plan: fsea_exec_u8_device, fsea_exec_u8_tiled_device, fsea_exec_u8_host, the history
 * ring, the gate, and -- since round 5 -- the frequency-shifted entry points (fsea_exec_u8_shifted_*: x[n] = (-1)^n w[n]
 * ((u8/256) e^{i phi} + 0.5 (1 + i)), kernels `*_u8_rot_win`) and the f64-input one (fsea_exec_f64_host: x[n] = (-1)^n
 * w[n] f64[n], src/nrf.c:607-612 with the taper beside the sign, kernels `*_f32_win`), i.e. the whole
 * nrf_freq_shifter -> nrf_fft chain of lua/fft-shifted.lua:52-55.  fsea_plan_set_window fails with FSEA_EINVAL on a plan
 * whose size has no kernel of its own (not a power of two in [32, 16384]).  fsea_plan_kernel_name's pointer stays valid
 * across this call (both names live as long as the plan); what it points to is the name in use when it was asked.
 * Synchronous (waits for the device); not to be called while another thread is launching the plan.
 * Precision: the kernels transform w[n] (u8[n] - 128) and add the offset-binary DC term back as its known spectrum
 * (computed in double at this call) when that spectrum is confined to the bins around n/2 -- every cosine-sum
 * taper (Hann, Hamming, Blackman, Blackman-Harris, flat-top; fsea_plan_window_form() == 1; what is left out is the
 * f32 rounding of the weights themselves, ~2e-8 of the DC term's rms).  Any other w (Kaiser, a truncated Gaussian,
 * arbitrary values) is applied to the offset-binary values themselves (form 2): same result, f32 rounding noise
 * relative to the DC term instead of to the signal (~1.2e-7 of the DC term's rms in every bin; inside the stated
 * tolerance either way). */
int fsea_plan_set_window(fsea_plan *plan, const float *w);
/* 0 = no window, 1 = centred form, 2 = offset-binary form (see above). */
int fsea_plan_window_form(const fsea_plan *plan);

/* Standard tapers in their periodic ("DFT-even") form, w[j] = sum_k (-1)^k a_k cos(2 pi k j / n) -- what
 * scipy.signal.get_window(name, n) returns -- evaluated in double, rounded to float. */
enum {
    FSEA_WINDOW_RECT = 0,
    FSEA_WINDOW_HANN = 1,            /* 0.5, 0.5 */
    FSEA_WINDOW_HAMMING = 2,         /* 0.54, 0.46 */
    FSEA_WINDOW_BLACKMAN = 3,        /* 0.42, 0.5, 0.08 */
    FSEA_WINDOW_BLACKMANHARRIS = 4,  /* 0.35875, 0.48829, 0.14128, 0.01168 */
    FSEA_WINDOW_FLATTOP = 5          /* 0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368 */
};
int fsea_window_fill(int kind, int n, float *w);

/* Launch geometry the plan would use for n_frames (persistent grid, workgroup
 * size, static LDS bytes per workgroup).  Any out-pointer may be NULL. */
int fsea_plan_grid(const fsea_plan *plan, size_t n_frames, unsigned *grid, unsigned *block,
                   size_t *lds_bytes);

/* Bytes of one output row for the plan's mode (fft_size * element size). */
size_t fsea_plan_row_bytes(const fsea_plan *plan);
int fsea_plan_fft_size(const fsea_plan *plan);

/* How a launch's frames are handed to the persistent workgroups of the sizes whose frames span several wavefronts
 * (4096 points and up): FSEA_UNITS_TICKETS = atomic ticket pools per XCD with stealing (evens out unequal progress;
 * best from more than 16 units -- frames, pairs of frames at 4096 points -- per workgroup, which is where AUTO
 * switches), FSEA_UNITS_STATIC = unit k of workgroup b is b + k * grid (no atomics;
 * best for short launches), FSEA_UNITS_AUTO (default) = chosen per launch by its length.  Results are identical; the
 * setting exists for measurements and tests.  Takes effect from the next launch; not to be changed while another
 * thread is launching the plan. */
#define FSEA_UNITS_AUTO 0
#define FSEA_UNITS_STATIC 1
#define FSEA_UNITS_TICKETS 2
int fsea_plan_set_unit_distribution(fsea_plan *plan, int policy);

/* Device-resident execution.  d_iq: device pointer to interleaved 8-bit IQ,
 * at least 2*((n_frames-1)*hop + fft_size) bytes, 16-byte aligned.
 * flip != 0: bytes are raw HackRF int8 and the kernel applies b ^ 0x80
 * (src/nrf.c:103-106); flip == 0: bytes are already offset-binary (RTL-SDR).
 * d_out: device pointer, n_frames rows of fsea_plan_row_bytes().
 * stream: hipStream_t as void*; NULL is HIP's null (default) stream, which is
 * also what torch.cuda.current_stream().cuda_stream is unless a side stream
 * is current.  Asynchronous with respect to the host.
 * A plan may be launched from several host threads and on any number of streams; launches on
 * one stream run in order, launches on different streams may overlap (up to 64 long launches in flight at a time;
 * a 65th waits for the oldest -- inside the call, holding the plan's slot table, so that other threads launching
 * the same plan wait with it).  The calls leave the caller's current HIP device unchanged.
 * The call enqueues one kernel; a long launch of the multi-wave sizes (4096 points and up, frames handed out by the
 * ticket pools) additionally records an event behind it, by which its counter slot is recycled.  While the stream is
 * being captured into a hipGraph nothing but the kernel is enqueued
 * (launch-bound consumers: tests/test_gpu_parity.py::test_launches_can_be_captured_into_a_hip_graph); a captured
 * launch keeps the ticket-counter slot of the stream it was captured on: replay one instance of such a graph at a
 * time, on the stream it was captured on or in order with that stream's other launches of the plan.  The slot stays
 * reserved for that stream until fsea_plan_reset (the graph's kernel node holds its address; the stream's own un-captured
 * launches go on using it): at most 64 distinct streams may hold captured launches of one plan at a time
 * (fsea_plan_release_stream gives one stream's slot back once its graphs are destroyed; fsea_plan_reset releases them all
 * and invalidates the graphs).  Plans of the sizes without a
 * kernel of their own (Bluestein / four-step) refuse a capturing stream with FSEA_EINVAL. */
int fsea_exec_u8_device(fsea_plan *plan, const void *d_iq, size_t n_frames, int flip,
                        void *d_out, void *stream);

/* The same transform with the rows written straight into a stitched image: the launch's frames are
 * consecutive `tile_rows`-row tiles (frame f = row f % tile_rows of tile f / tile_rows), and tile k lands
 * at columns [first_x + k*tile_step, ... + fft_size) of rows 0..tile_rows-1 of d_image (image_rows rows of
 * image_stride elements of the plan's output type).  Tiles are WRITTEN, not max-composited, so tile_step
 * must be >= fft_size; on a zeroed image that equals img_gray_copy's max() of c/fft-stitch-broad.c:62-87
 * (WIDTH_STEP == FFT_SIZE there), without the tile stack, the image read and the second pass.  Overlapping
 * tiles (c/fft-stitch.c, step < fft_size) go through fsea_exec_u8_device + fsea_stitch_tiles_device.
 * n_frames must be whole tiles; tile_rows a multiple of the size's frames per workgroup (1 for
 * fft_size >= 8192, 2 at 4096, ... 64 at 32: powers of two up to 64 always work); image_stride, first_x,
 * tile_step multiples of 4 elements; d_image 16-byte aligned. */
int fsea_exec_u8_tiled_device(fsea_plan *plan, const void *d_iq, size_t n_frames, int flip, void *d_image,
                              size_t image_rows, size_t image_stride, size_t first_x, size_t tile_rows,
                              size_t tile_step, void *stream);

/* Host-buffer execution; returns when `out` is complete.  Batches of up to 256 KiB (in + out: one nrf_fft_process
 * row) run as ONE launch on device-mapped pinned staging.  Larger ones are pipelined in chunks of whole frames: chunk
 * c travels to the device while chunk c-1 is transformed and the rows of chunk c-2 travel back, on three streams --
 * the streaming shape of the reference's tools (c/fft-batch.c:54-102: a transfer in, a row out) at the granularity a
 * PCIe link wants.  `iq` and `out` may be any host memory; pages that are not pinned yet are pinned in place for the
 * duration of the call (hipHostRegister) so that the two directions overlap, and memory from fsea_host_alloc (or
 * hipHostMalloc) skips that step.  Link-bound: 64 MiB in + 128 MiB out in 2.6-2.9 ms on an MI355X host
 * (profiles/r03_host_path.txt).  The same holds for the _shifted_ and _f64_ forms below. */
int fsea_exec_u8_host(fsea_plan *plan, const uint8_t *iq, size_t n_frames, int flip,
                      void *out);

/* Frequency-shifted spectrum, the device-resident form of
 *   nrf_freq_shifter_process(shifter, samples); nrf_fft_process(fft, shifter_buffer)
 * (src/nrf.c:843-866, 598-631; lua/fft-shifted.lua:52-55) with the rotation fused into the FFT
 * kernel's load.  Stream sample m (frame f, sample n: m = f*hop + n) becomes
 *   y[m] = (u8[m] / 256) * e^{+2 pi i (phase0_cycles + m * cycles_per_sample)} + 0.5 (1 + i)
 * and frame f transforms x[n] = (-1)^n y[f*hop + n]; epilogue per the plan's mode.
 * cycles_per_sample = freq_offset / sample_rate; phase0_cycles continues a stream across calls
 * (a shifter that has consumed M samples is at phase M * cycles_per_sample).  Other arguments
 * as fsea_exec_u8_device / fsea_exec_u8_host. */
int fsea_exec_u8_shifted_device(fsea_plan *plan, const void *d_iq, size_t n_frames, int flip,
                                double cycles_per_sample, double phase0_cycles, void *d_out,
                                void *stream);
int fsea_exec_u8_shifted_host(fsea_plan *plan, const uint8_t *iq, size_t n_frames, int flip,
                              double cycles_per_sample, double phase0_cycles, void *out);

/* F64 interleaved complex input (src/nrf.c:607-609: no /256, no flip). */
int fsea_exec_f64_host(fsea_plan *plan, const double *iq, size_t n_frames, void *out);

/* Device-resident history of MAG_F32 rows, newest first: nrf_fft's `buffer` (src/nrf.h:133,
 * src/nrf.c:565, 616-617) kept in HBM as a ring.  push = one nrf_fft_process: the kernel writes the
 * new row straight into the ring (the reference memmoves the whole history by one row first);
 * shift = nrf_fft_shift's per-row scroll for an integer number of bins (src/nrf.c:569-596: > 0 moves
 * rows left, < 0 right, vacated bins zero, |shift| >= fft_size clears the history), as a kernel;
 * get = nrf_fft_get_buffer: ONE device-to-host transfer of rows * fft_size f32 and one widening to
 * f64 into `out` (rows * fft_size doubles).  The plan must be a MAG_F32 plan and must outlive every
 * push / shift / get on the history (destroy the history first; fsea_history_destroy itself no longer touches the
 * plan).  All calls are synchronous. */
int fsea_history_create(fsea_plan *plan, int rows, fsea_history **history);
int fsea_history_destroy(fsea_history *history);
int fsea_history_push_u8_host(fsea_history *history, const uint8_t *iq, int flip);
int fsea_history_push_f64_host(fsea_history *history, const double *iq);
int fsea_history_shift(fsea_history *history, int shift);
int fsea_history_get_f64(fsea_history *history, double *out);

/* Mean of sqrt(re^2+im^2) over the first n_frames rows of a device-resident
 * u8 IQ block (c/fft-batch-broad.c:81-98).  Synchronous. */
int fsea_mean_magnitude_u8_device(fsea_plan *plan, const void *d_iq, size_t n_frames,
                                  int flip, double *mean, void *stream);

/* dst[dst_y + y][dst_x + j] = max(dst, src[y][j]) for a width x height u8 tile
 * (c/fft-stitch.c:46-54).  dst is dst_height rows of dst_stride pixels; a tile that does not lie
 * inside it is rejected with FSEA_EINVAL (nothing is written).  Device pointers; asynchronous on
 * `stream`. */
int fsea_composite_max_device(void *d_dst, const void *d_src, uint32_t dst_x,
                              uint32_t dst_y, uint32_t width, uint32_t height,
                              uint32_t dst_stride, uint32_t dst_height, uint32_t src_stride, int device,
                              void *stream);

/* The whole stitch loop of c/fft-stitch*.c:167-189 for a contiguous stack of tiles
 * ([k][y][x], each width x height): tile k is max-composited at x = first_x +
 * k * width_step.  Overlapping neighbours (width_step < width) are processed in
 * separate race-free launches.  Device pointers; asynchronous on `stream`. */
int fsea_stitch_tiles_device(void *d_image, const void *d_tiles, uint32_t n_tiles, uint32_t first_x,
                             uint32_t width_step, uint32_t width, uint32_t height,
                             uint32_t image_stride, int device, void *stream);

/* Pinned (page-locked) host memory for the buffers handed to the *_host entry points, so that C callers need no HIP
 * headers: copies from and to it are asynchronous without a per-call registration.  Free with fsea_host_free. */
int fsea_host_alloc(size_t bytes, void **ptr);
int fsea_host_free(void *ptr);

/* Small device-memory helpers so that C callers need no HIP headers. */
int fsea_device_alloc(int device, size_t bytes, void **d_ptr);
int fsea_device_free(int device, void *d_ptr);
int fsea_copy_to_device(int device, void *d_dst, const void *src, size_t bytes);
int fsea_copy_to_host(int device, void *dst, const void *d_src, size_t bytes);
/* Waits for `stream` (NULL = the null stream) on the plan's device. */
int fsea_stream_synchronize(fsea_plan *plan, void *stream);

/* Streams and asynchronous copies for C callers (a hipStream_t as void*, non-blocking with respect to the null stream).
 * A consumer of independent batches -- captures of a sweep, blocks of a stream -- submits consecutive batches
 * ALTERNATELY ON TWO STREAMS with double-buffered device memory: upload, transform and download of batch k + 1 overlap
 * the drain of batch k, and the ramp and tail of each launch (a few microseconds in which the chip is not full) are
 * covered by its neighbour: +8...12 % on resident batches (bench.py: extra.two_stream_*), more when copies are in the
 * loop.  INTEGRATION.md section 2 shows the pattern; fsea-fft-batch and fsea-fft-sweep use it.  Host memory should come
 * from fsea_host_alloc (pinned), or the copies fall back to staged, synchronous ones. */
int fsea_stream_create(int device, void **stream);
int fsea_stream_destroy(int device, void *stream);
int fsea_copy_to_device_async(int device, void *d_dst, const void *src, size_t bytes, void *stream);
int fsea_copy_to_host_async(int device, void *dst, const void *d_src, size_t bytes, void *stream);

/* Recovery after an aborted launch (device fault, killed context): waits for the device and
 * re-zeroes the plan's internal frame-distribution counters.  Not needed in normal operation. */
int fsea_plan_reset(fsea_plan *plan);

/* Gives back the frame-distribution counter slot `stream` holds in this plan, in particular one reserved by a captured
 * launch: call it once the hipGraphs captured on that stream are destroyed (before destroying the stream), so that an
 * application capturing on short-lived streams does not run out of the 64 slots.  Waits for the stream's last un-captured
 * launch of the plan.  FSEA_OK also when the stream holds no slot; FSEA_EINVAL while the stream is capturing. */
int fsea_plan_release_stream(fsea_plan *plan, void *stream);

/* Name of the kernel the plan launches for raw int8 input (flip != 0), for matching rocprof rows:
 * MAG_F32, DB5_U8_DCFIX and DB10_U8 plans have compile-time kernels (`*_u8_mag`, `*_u8_db5`,
 * `*_u8_db10`); the other modes, and any mode with flip == 0, run the run-time-mode kernel `*_u8`.
 * Kernel variants, per-workgroup traces and the timing helper live in the separate tuning
 * library (include/fsea_tune.h, libfsea_hip_tune.so); this library has one kernel set per size. */
const char *fsea_plan_kernel_name(const fsea_plan *plan);

const char *fsea_last_error_string(void);

#ifdef __cplusplus
}
#endif
#endif
