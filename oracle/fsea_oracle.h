/*
 * fsea_oracle.h -- CPU restatement (C99, double precision) of the frequensea
 * IQ-FFT spectrum path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this library.  The product (libfsea_hip.so / libfsea_nrf.so)
 * never does: it has no CPU fallback and fails loudly without a GPU.
 *
 * PARITY PINNING STATUS: "parity unpinned by the reference".  The reference
 * ships no tests / golden vectors and its FFT arithmetic
 * lives in FFTW3 (double precision, version unpinned: CMakeLists.txt:16,
 * README.md:19,23), which is absent from /root/reference and from this
 * image; src/nrf.c cannot be compiled here without stand-ins for
 * fftw3.h / libhackrf / rtl-sdr / OpenAL headers, so it is treated as
 * unbuildable.  What pins this oracle instead:
 *   - FFTW's published definition of fftw_plan_dft_1d(FFTW_FORWARD):
 *     unnormalised X[k] = sum_j x[j] exp(-2 pi i j k / N);
 *   - an O(N^2) long-double DFT in this file (orc_dft_naive);
 *   - scipy/pocketfft on the reference's recorded rfdata captures
 *     (tests/golden/make_golden.py, committed fixtures);
 *   - the known-answer values SURVEY.md section 8(c) recorded from the
 *     reference's own nrf.c;
 *   - an FFTW3-API library driven through the reference's own calls
 *     (fftw_plan_dft_1d + fftw_execute; Intel MKL's interface in this image,
 *     orc_time_mag_rows_fftw + tests/test_oracle.py) -- an independent
 *     implementation, not FFTW itself;
 *   - oracle/_ref/libnut_ref.so = the reference's src/nut.c compiled as is
 *     (it needs no third-party headers) pins the nut_buffer conventions;
 *   - oracle/_ref/fft-stitch-broad = the reference's c/fft-stitch-broad.c
 *     compiled as is against the image's libpng pins orc_composite_max and
 *     the PNG tile format (tests/test_reference_tools.py).
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference).
 */
#ifndef FSEA_ORACLE_H
#define FSEA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* a1: HackRF int8 -> offset binary.  src/nrf.c:100-109, c/fft-batch.c:63-64 */
void orc_flip_u8(const uint8_t *in, uint8_t *out, size_t n_bytes);

/* a4: u8 interleaved IQ -> complex double with (-1)^n centring.
 * src/nrf.c:601-614 (U8 branch: u8/256.0). out has 2*n_samples doubles. */
void orc_unpack_center_u8(const uint8_t *iq, size_t n_samples, double *out);

/* a4, F64 branch: src/nrf.c:607-612. */
void orc_unpack_center_f64(const double *iq, size_t n_samples, double *out);

/* a5: unnormalised forward DFT, -1 exponent (what fftw_execute does for the
 * plan made at src/nrf.c:564 / c/fft-batch.c:144).  n must be a power of 2.
 * in/out: interleaved re,im doubles; out-of-place. Returns 0, or -1 on bad n. */
int orc_fft_forward(const double *in, double *out, int n);

/* Same transform as an O(n^2) long-double sum; any n >= 1.  Self-check only. */
void orc_dft_naive(const double *in, double *out, int n);

/* a7: magnitude row + DC compensation.  src/nrf.c:619-630.
 * row[i] = sqrt(re^2+im^2); row[n/2] = row[n/2-1]. */
void orc_mag_row(const double *spectrum, int n, double *row);

/* a6: history scroll (rows 0..h-2 -> 1..h-1), defined as memmove.
 * src/nrf.c:616-617. */
void orc_history_scroll(double *history, int n, int h);

/* a9: nrf_fft_shift.  src/nrf.c:569-596. */
void orc_fft_shift(double *history, int n, int h, double d);

/* a11/a12: dB pixel row.  c/fft-batch.c:83-94 (scale 10, dcfix 0),
 * c/fft-batch-broad.c:106-121 (scale 5, dcfix 1). */
void orc_db_u8_row(const double *spectrum, int n, double scale, int dcfix,
                   uint8_t *row);

/* a12: mean magnitude over count complex bins. c/fft-batch-broad.c:81-98.
 * The reference skips the frequency when the mean is < 1.1. */
double orc_mean_magnitude(const double *spectrum, size_t count);

/* a14: max-composite of a w x h tile into dst at (dst_x, dst_y).
 * c/fft-stitch.c:46-54, c/fft-stitch-broad.c:28-36. */
void orc_composite_max(uint8_t *dst, const uint8_t *src, uint32_t dst_x,
                       uint32_t dst_y, uint32_t src_x, uint32_t src_y,
                       uint32_t width, uint32_t height, uint32_t dst_stride,
                       uint32_t src_stride);

/* Whole-row pipelines (flip? -> unpack -> FFT -> epilogue), frame f starts at
 * sample f*hop of `iq`.  mode: 0 = MAG (a7), 1 = DB10_U8 (a11),
 * 2 = DB5_U8_DCFIX (a12), 3 = complex spectrum (a5 only, 2n doubles/row),
 * 4 = MAG without the DC patch, 5 = 10*log10(pwr + 1e-20) as double.
 * out is double[n_frames*n] for modes 0,4,5, uint8_t[n_frames*n] for 1,2,
 * double[n_frames*2n] for 3.  Returns 0 on success. */
int orc_rows(const uint8_t *iq, size_t n_frames, int n, size_t hop, int flip,
             int mode, void *out);

/* orc_rows with a taper window in the weight slot of the unpack loop.  The reference's loop (src/nrf.c:601-614;
 * c/fft-batch.c:62-68) weights sample ii by powf(-1, ii) alone; this is the same loop with w[ii] beside it:
 *   x[ii] = (-1)^ii * window[ii] * (u8[ii] / 256.0)
 * (window == NULL or all ones: orc_rows itself).  Everything behind the unpack -- transform, magnitude and its
 * bin n/2 := bin n/2-1 copy, dB pixels -- is the reference's, unchanged.  An EXTENSION named by BASELINE.json
 * (north_star: "fused unpack+window prologue"; config 5's STFT), not reference behaviour: the reference has no taper
 * (SURVEY.md section 0). */
int orc_rows_windowed(const uint8_t *iq, size_t n_frames, int n, size_t hop, int flip, int mode,
                      const double *window, void *out);

/* The periodic ("DFT-even") cosine-sum tapers, w[j] = sum_k (-1)^k a_k cos(2 pi k j / n): kind 0 rectangular, 1 Hann
 * (0.5, 0.5), 2 Hamming (0.54, 0.46), 3 Blackman (0.42, 0.5, 0.08), 4 Blackman-Harris, 5 flat-top -- the coefficient
 * sets of scipy.signal.windows (tests/test_oracle.py checks them against scipy.signal.get_window). */
int orc_window_fill(int kind, int n, double *w);

/* orc_rows / orc_rows_windowed (window == NULL: the reference's rectangular frames) with the frames cut into n_threads
 * contiguous ranges, one pthread each running the single-threaded function on its range -- the same rows bit for bit, in
 * the time the GPU tier can afford for EVERY row of a BASELINE.json configuration (4096 x 8192, 32768 x 1024,
 * 131072 x 4096 pixels, 32767 x 16384).  Restates src/nrf.c:598-631 per frame like orc_rows. */
int orc_rows_mt(const uint8_t *iq, size_t n_frames, int n, size_t hop, int flip, int mode,
                const double *window, int n_threads, void *out);

/* nrf_freq_shifter_process on interleaved IQ (src/nrf.c:843-866): for every sample,
 *   out_i = vi*cos - vq*sin + 0.5,  out_q = vi*sin + vq*cos + 0.5,
 * then (cos, sin) advance by the angle 2*pi*freq_offset/sample_rate through the reference's own
 * recurrence (new_sin = cos*dsin + sin*dcos; new_cos = cos*dcos - sin*dsin, in double).
 * vi/vq are nut_buffer_get_f64 values: u8/256.0 for `iq_u8`, or taken from `iq_f64` when
 * iq_u8 is NULL.  *cosine / *sine carry the state across calls (1, 0 initially). */
void orc_freq_shift(const uint8_t *iq_u8, const double *iq_f64, size_t n_samples, int freq_offset,
                    int sample_rate, double *cosine, double *sine, double *out);

/* orc_rows for a frequency-shifted stream: frame f transforms
 *   x[n] = (-1)^n * ( (u8[m]/256) * e^{+2 pi i (phase0 + m*cycles_per_sample)} + 0.5 (1+i) ),
 *   m = f*hop + n,
 * which is nrf_freq_shifter_process followed by nrf_fft_process' F64 branch (src/nrf.c:843-866,
 * 607-612) with the phase in closed form instead of the recurrence (they agree to ~1e-12 over
 * 2^17 samples; tests/test_oracle.py checks it).  Modes and `out` as orc_rows. */
int orc_rows_shifted(const uint8_t *iq, size_t n_frames, int n, size_t hop, int flip, int mode,
                     double cycles_per_sample, double phase0_cycles, void *out);

/* orc_rows_shifted with a taper beside the (-1)^n of the F64 branch (window == NULL: orc_rows_shifted itself):
 *   x[n] = (-1)^n * window[n] * ( (u8[m]/256) * e^{+2 pi i (...)} + 0.5 (1+i) )
 * -- the nrf_freq_shifter -> nrf_fft chain of lua/fft-shifted.lua:52-55 on a plan with a window.  An extension, as
 * orc_rows_windowed. */
int orc_rows_shifted_windowed(const uint8_t *iq, size_t n_frames, int n, size_t hop, int flip, int mode,
                              double cycles_per_sample, double phase0_cycles, const double *window, void *out);

/* nrf_fft_process' F64 branch (src/nrf.c:607-612: x[ii] = (-1)^ii * f64) on whole rows, with an optional taper beside the
 * sign (window == NULL: the reference): frame f = complex samples [f*hop, f*hop + n) of iq (interleaved doubles). */
int orc_rows_f64(const double *iq, size_t n_frames, int n, size_t hop, int mode, const double *window, void *out);

/* The reference-shaped per-frame loop used as bench.py's cpu_baseline
 * ("port"): a1 flip -> a4 unpack (n samples) -> a5 FFT -> a7 magnitude, on
 * one core, with a pre-planned twiddle table.  Returns seconds elapsed for
 * n_frames frames (monotonic clock); sink receives the last row. */
double orc_time_mag_rows(const uint8_t *iq, size_t n_frames, int n, size_t hop,
                         double *sink);

/* Same loop with frames sharded over n_threads pthreads, one plan per thread
 * (BASELINE.md section 4 item 2(ii)).  Returns wall seconds of the transform loop (thread start-up
 * and planning excluded). */
double orc_time_mag_rows_mt(const uint8_t *iq, size_t n_frames, int n, size_t hop,
                            int n_threads, double *checksum);

/* The same reference-shaped loop with the transform done by an FFTW3-API library loaded at run
 * time -- exactly the calls the reference makes (src/nrf.c:562-564,615: fftw_plan_dft_1d(n, in, out,
 * FFTW_FORWARD, FFTW_MEASURE), fftw_execute), one plan per thread, planner calls serialised.
 * `lib`: path or soname, e.g. "libfftw3.so.3", or Intel MKL's "libmkl_rt.so", which exports the
 * FFTW3 interface.  If rows != NULL the first min(n_frames, rows_cap) magnitude rows are stored
 * there (rows_cap * n doubles) so that the library can be checked against the oracle's own FFT.
 * Every thread walks its share of the frames `passes` times; threads and plans are made once,
 * before the timed region (planning is a one-off cost in the reference too, src/nrf.c:564).
 * Returns wall seconds, or a negative value when the library or a symbol cannot be loaded. */
double orc_time_mag_rows_fftw(const char *lib, const uint8_t *iq, size_t n_frames, int n, size_t hop,
                              int n_threads, int passes, double *rows, size_t rows_cap, double *checksum);

#ifdef __cplusplus
}
#endif
#endif
