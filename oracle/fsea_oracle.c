/*
 * fsea_oracle.c -- CPU restatement of the frequensea IQ-FFT path (C99, f64).
 * TEST INFRASTRUCTURE ONLY; see fsea_oracle.h for who may call it and for the
 * parity-pinning status ("parity unpinned by the reference": no reference
 * tests exist and FFTW is absent; pinned by definition + independent DFTs).
 *
 * Citations are path:line under /root/reference.
 */
#define _POSIX_C_SOURCE 200809L
#include "fsea_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- a1 ------------------------------------------------------------- */

/* src/nrf.c:100-109: u8i = (u8i + 128) % 256 for HackRF/dummy devices. */
void orc_flip_u8(const uint8_t *in, uint8_t *out, size_t n_bytes) {
    for (size_t i = 0; i < n_bytes; i++) {
        out[i] = (uint8_t)((in[i] + 128) % 256);
    }
}

/* ---- a4 ------------------------------------------------------------- */

/* src/nrf.c:601-614.  powf(-1, ii) is exactly +1 for even ii, -1 for odd. */
void orc_unpack_center_u8(const uint8_t *iq, size_t n_samples, double *out) {
    for (size_t ii = 0; ii < n_samples; ii++) {
        double di = iq[2 * ii] / 256.0;
        double dq = iq[2 * ii + 1] / 256.0;
        double sign = (ii & 1) ? -1.0 : 1.0;
        out[2 * ii] = sign * di;
        out[2 * ii + 1] = sign * dq;
    }
}

void orc_unpack_center_f64(const double *iq, size_t n_samples, double *out) {
    for (size_t ii = 0; ii < n_samples; ii++) {
        double sign = (ii & 1) ? -1.0 : 1.0;
        out[2 * ii] = sign * iq[2 * ii];
        out[2 * ii + 1] = sign * iq[2 * ii + 1];
    }
}

/* nrf_fft_process' F64 branch (src/nrf.c:607-612) on whole rows, with an optional taper beside the (-1)^ii (window NULL:
 * the reference): frame f = complex samples [f*hop, f*hop + n) of iq (interleaved doubles).  Modes and out as orc_rows. */
int orc_rows_f64(const double *iq, size_t n_frames, int n, size_t hop, int mode, const double *window, void *out);

/* ---- a5 ------------------------------------------------------------- */

static int ilog2_exact(int n) {
    int l = 0;
    if (n < 1) return -1;
    while ((1 << l) < n) l++;
    return ((1 << l) == n) ? l : -1;
}

/* Twiddle table W[k] = exp(-2 pi i k / n), k < n/2, evaluated in long double
 * and rounded once, so that table error stays at 0.5 ulp. */
typedef struct {
    int n;
    int log2n;
    double *w;        /* n/2 complex */
    uint32_t *bitrev; /* n entries */
} orc_plan;

static int orc_plan_init(orc_plan *p, int n) {
    int l = ilog2_exact(n);
    if (l < 0) return -1;
    p->n = n;
    p->log2n = l;
    p->w = (double *)malloc(sizeof(double) * (size_t)(n > 1 ? n : 2));
    p->bitrev = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
    if (!p->w || !p->bitrev) return -1;
    for (int k = 0; k < n / 2; k++) {
        long double a = -2.0L * 3.14159265358979323846264338327950288L *
                        (long double)k / (long double)n;
        p->w[2 * k] = (double)cosl(a);
        p->w[2 * k + 1] = (double)sinl(a);
    }
    for (int i = 0; i < n; i++) {
        uint32_t r = 0;
        for (int b = 0; b < l; b++) {
            if (i & (1 << b)) r |= 1u << (l - 1 - b);
        }
        p->bitrev[i] = r;
    }
    return 0;
}

static void orc_plan_free(orc_plan *p) {
    free(p->w);
    free(p->bitrev);
    p->w = NULL;
    p->bitrev = NULL;
}

/* Decimation-in-time radix-2, bit-reversed load, natural-order output. */
static void orc_plan_exec(const orc_plan *p, const double *in, double *out) {
    const int n = p->n;
    for (int i = 0; i < n; i++) {
        uint32_t r = p->bitrev[i];
        out[2 * r] = in[2 * i];
        out[2 * r + 1] = in[2 * i + 1];
    }
    for (int half = 1; half < n; half <<= 1) {
        const int step = n / (2 * half); /* twiddle stride in the n/2 table */
        for (int base = 0; base < n; base += 2 * half) {
            for (int k = 0; k < half; k++) {
                const double wr = p->w[2 * (k * step)];
                const double wi = p->w[2 * (k * step) + 1];
                double *a = out + 2 * (base + k);
                double *b = out + 2 * (base + k + half);
                const double tr = wr * b[0] - wi * b[1];
                const double ti = wr * b[1] + wi * b[0];
                b[0] = a[0] - tr;
                b[1] = a[1] - ti;
                a[0] = a[0] + tr;
                a[1] = a[1] + ti;
            }
        }
    }
}

int orc_fft_forward(const double *in, double *out, int n) {
    orc_plan p;
    if (orc_plan_init(&p, n) != 0) return -1;
    orc_plan_exec(&p, in, out);
    orc_plan_free(&p);
    return 0;
}

void orc_dft_naive(const double *in, double *out, int n) {
    const long double tau = 2.0L * 3.14159265358979323846264338327950288L;
    for (int k = 0; k < n; k++) {
        long double sr = 0.0L, si = 0.0L;
        for (int j = 0; j < n; j++) {
            /* reduce j*k mod n first so the angle stays small and exact */
            long long m = ((long long)j * (long long)k) % n;
            long double a = -tau * (long double)m / (long double)n;
            long double c = cosl(a), s = sinl(a);
            sr += (long double)in[2 * j] * c - (long double)in[2 * j + 1] * s;
            si += (long double)in[2 * j] * s + (long double)in[2 * j + 1] * c;
        }
        out[2 * k] = (double)sr;
        out[2 * k + 1] = (double)si;
    }
}

/* ---- a7 ------------------------------------------------------------- */

/* src/nrf.c:619-630.  Bin n/2 takes the already computed left neighbour. */
void orc_mag_row(const double *spectrum, int n, double *row) {
    for (int i = 0; i < n; i++) {
        if (i == n / 2 && i > 0) {
            row[i] = row[i - 1];
        } else {
            double fi = spectrum[2 * i];
            double fq = spectrum[2 * i + 1];
            row[i] = sqrt(fi * fi + fq * fq);
        }
    }
}

/* ---- a6 / a9 ---------------------------------------------------------- */

/* src/nrf.c:617 (overlapping memcpy in the reference; defined as memmove). */
void orc_history_scroll(double *history, int n, int h) {
    if (h > 1) {
        memmove(history + n, history, sizeof(double) * (size_t)n * (size_t)(h - 1));
    }
}

/* src/nrf.c:569-596. */
void orc_fft_shift(double *history, int n, int h, double d) {
    int shift_pixels = (int)round(n / d);
    if (shift_pixels == 0) {
        return;
    } else if (abs(shift_pixels) >= n) {
        memset(history, 0, sizeof(double) * (size_t)n * (size_t)h);
    } else {
        for (int y = 0; y < h; y++) {
            double *row = history + (size_t)y * (size_t)n;
            if (shift_pixels > 0) {
                for (int x = 0; x < n - shift_pixels; x++) row[x] = row[x + shift_pixels];
                for (int x = n - shift_pixels; x < n; x++) row[x] = 0;
            } else {
                for (int x = n - 1; x >= -shift_pixels; x--) row[x] = row[x + shift_pixels];
                for (int x = 0; x < -shift_pixels; x++) row[x] = 0;
            }
        }
    }
}

/* ---- a11 / a12 -------------------------------------------------------- */

/* c/fft-batch.c:35-37: clamp_u8(int v, min, max); the double -> int
 * conversion at the call site truncates toward zero (C semantics). */
static uint8_t orc_clamp_u8(int v, uint8_t lo, uint8_t hi) {
    return (uint8_t)(v < lo ? lo : v > hi ? hi : v);
}

void orc_db_u8_row(const double *spectrum, int n, double scale, int dcfix,
                   uint8_t *row) {
    for (int x = 0; x < n; x++) {
        double ci = spectrum[2 * x];
        double cq = spectrum[2 * x + 1];
        double pwr = ci * ci + cq * cq;
        double pwr_dbfs = 10.0 * log10(pwr + 1.0e-20);
        pwr_dbfs = pwr_dbfs * scale;
        uint8_t v = orc_clamp_u8((int)pwr_dbfs, 0, 255);
        if (dcfix && x == n / 2 && x > 0) {
            v = row[x - 1]; /* c/fft-batch-broad.c:114-116 */
        }
        row[x] = v;
    }
}

double orc_mean_magnitude(const double *spectrum, size_t count) {
    double total = 0;
    for (size_t i = 0; i < count; i++) {
        double ci = spectrum[2 * i];
        double cq = spectrum[2 * i + 1];
        total += sqrt(ci * ci + cq * cq);
    }
    return total / (double)count;
}

/* ---- a14 -------------------------------------------------------------- */

void orc_composite_max(uint8_t *dst, const uint8_t *src, uint32_t dst_x,
                       uint32_t dst_y, uint32_t src_x, uint32_t src_y,
                       uint32_t width, uint32_t height, uint32_t dst_stride,
                       uint32_t src_stride) {
    for (uint32_t i = 0; i < height; i++) {
        for (uint32_t j = 0; j < width; j++) {
            size_t d = (size_t)(dst_y + i) * dst_stride + dst_x + j;
            size_t s = (size_t)(src_y + i) * src_stride + src_x + j;
            dst[d] = dst[d] > src[s] ? dst[d] : src[s];
        }
    }
}

/* ---- whole rows --------------------------------------------------------- */

static void orc_epilogue(int n, int mode, const double *tmp_out, void *out_row);

static void orc_one_row(const orc_plan *p, const uint8_t *iq, int flip, int mode,
                        uint8_t *tmp_u8, double *tmp_in, double *tmp_out,
                        void *out_row) {
    const int n = p->n;
    const uint8_t *src = iq;
    if (flip) {
        orc_flip_u8(iq, tmp_u8, (size_t)2 * (size_t)n);
        src = tmp_u8;
    }
    orc_unpack_center_u8(src, (size_t)n, tmp_in);
    orc_plan_exec(p, tmp_in, tmp_out);
    orc_epilogue(n, mode, tmp_out, out_row);
}

static void orc_epilogue(int n, int mode, const double *tmp_out, void *out_row) {
    switch (mode) {
    case 0:
        orc_mag_row(tmp_out, n, (double *)out_row);
        break;
    case 1:
        orc_db_u8_row(tmp_out, n, 10.0, 0, (uint8_t *)out_row);
        break;
    case 2:
        orc_db_u8_row(tmp_out, n, 5.0, 1, (uint8_t *)out_row);
        break;
    case 3:
        memcpy(out_row, tmp_out, sizeof(double) * 2 * (size_t)n);
        break;
    case 4: {
        double *row = (double *)out_row;
        for (int i = 0; i < n; i++) {
            row[i] = sqrt(tmp_out[2 * i] * tmp_out[2 * i] + tmp_out[2 * i + 1] * tmp_out[2 * i + 1]);
        }
        break;
    }
    case 5: {
        double *row = (double *)out_row;
        for (int i = 0; i < n; i++) {
            double pwr = tmp_out[2 * i] * tmp_out[2 * i] + tmp_out[2 * i + 1] * tmp_out[2 * i + 1];
            row[i] = 10.0 * log10(pwr + 1.0e-20);
        }
        break;
    }
    default:
        break;
    }
}

int orc_rows(const uint8_t *iq, size_t n_frames, int n, size_t hop, int flip,
             int mode, void *out) {
    orc_plan p;
    if (mode < 0 || mode > 5) return -2;
    if (orc_plan_init(&p, n) != 0) return -1;
    uint8_t *tmp_u8 = (uint8_t *)malloc((size_t)2 * (size_t)n);
    double *tmp_in = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *tmp_out = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    size_t row_bytes = (mode == 1 || mode == 2) ? (size_t)n
                       : (mode == 3)            ? sizeof(double) * 2 * (size_t)n
                                                : sizeof(double) * (size_t)n;
    for (size_t f = 0; f < n_frames; f++) {
        orc_one_row(&p, iq + 2 * f * hop, flip, mode, tmp_u8, tmp_in, tmp_out,
                    (uint8_t *)out + f * row_bytes);
    }
    free(tmp_u8);
    free(tmp_in);
    free(tmp_out);
    orc_plan_free(&p);
    return 0;
}

int orc_rows_f64(const double *iq, size_t n_frames, int n, size_t hop, int mode, const double *window, void *out) {
    orc_plan p;
    if (mode < 0 || mode > 5) return -2;
    if (orc_plan_init(&p, n) != 0) return -1;
    double *tmp_in = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *tmp_out = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    size_t row_bytes = (mode == 1 || mode == 2) ? (size_t)n
                       : (mode == 3)            ? sizeof(double) * 2 * (size_t)n
                                                : sizeof(double) * (size_t)n;
    for (size_t f = 0; f < n_frames; f++) {
        orc_unpack_center_f64(iq + 2 * f * hop, (size_t)n, tmp_in); /* src/nrf.c:607-612 */
        if (window != NULL) {
            for (int k = 0; k < n; k++) {
                tmp_in[2 * k] *= window[k];
                tmp_in[2 * k + 1] *= window[k];
            }
        }
        orc_plan_exec(&p, tmp_in, tmp_out);
        orc_epilogue(n, mode, tmp_out, (uint8_t *)out + f * row_bytes);
    }
    free(tmp_in);
    free(tmp_out);
    orc_plan_free(&p);
    return 0;
}

/* ---- taper window (extension: BASELINE.json north_star / config 5; the reference's only weight is (-1)^ii) ---- */

int orc_rows_windowed(const uint8_t *iq, size_t n_frames, int n, size_t hop, int flip, int mode,
                      const double *window, void *out) {
    orc_plan p;
    if (window == NULL) return orc_rows(iq, n_frames, n, hop, flip, mode, out);
    if (mode < 0 || mode > 5) return -2;
    if (orc_plan_init(&p, n) != 0) return -1;
    uint8_t *tmp_u8 = (uint8_t *)malloc((size_t)2 * (size_t)n);
    double *tmp_in = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *tmp_out = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    size_t row_bytes = (mode == 1 || mode == 2) ? (size_t)n
                       : (mode == 3)            ? sizeof(double) * 2 * (size_t)n
                                                : sizeof(double) * (size_t)n;
    for (size_t f = 0; f < n_frames; f++) {
        const uint8_t *src = iq + 2 * f * hop;
        if (flip) {
            orc_flip_u8(src, tmp_u8, (size_t)2 * (size_t)n); /* src/nrf.c:100-109 */
            src = tmp_u8;
        }
        orc_unpack_center_u8(src, (size_t)n, tmp_in);        /* src/nrf.c:601-614: (-1)^ii * u8/256.0 */
        for (int ii = 0; ii < n; ii++) {                      /* ... and the taper beside the (-1)^ii */
            tmp_in[2 * ii] *= window[ii];
            tmp_in[2 * ii + 1] *= window[ii];
        }
        orc_plan_exec(&p, tmp_in, tmp_out);
        orc_epilogue(n, mode, tmp_out, (uint8_t *)out + f * row_bytes);
    }
    free(tmp_u8);
    free(tmp_in);
    free(tmp_out);
    orc_plan_free(&p);
    return 0;
}

int orc_window_fill(int kind, int n, double *w) {
    static const double coef[6][5] = {{1.0, 0, 0, 0, 0},
                                      {0.5, 0.5, 0, 0, 0},
                                      {0.54, 0.46, 0, 0, 0},
                                      {0.42, 0.5, 0.08, 0, 0},
                                      {0.35875, 0.48829, 0.14128, 0.01168, 0},
                                      {0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368}};
    const double tau = 6.283185307179586476925286766559;
    if (kind < 0 || kind > 5 || n < 1) return -1;
    for (int j = 0; j < n; j++) {
        double v = 0.0, sign = 1.0;
        for (int k = 0; k < 5; k++) {
            if (coef[kind][k] != 0.0) v += sign * coef[kind][k] * cos(tau * (double)k * (double)j / (double)n);
            sign = -sign;
        }
        w[j] = v;
    }
    return 0;
}

/* ---- frequency shifter (src/nrf.c:843-866) ------------------------------- */

void orc_freq_shift(const uint8_t *iq_u8, const double *iq_f64, size_t n_samples, int freq_offset,
                    int sample_rate, double *cosine, double *sine, double *out) {
    const double tau = 6.283185307179586476925286766559;
    const double dcos = cos(tau * freq_offset / (double)sample_rate);
    const double dsin = sin(tau * freq_offset / (double)sample_rate);
    double c = *cosine, s = *sine;
    for (size_t k = 0; k < n_samples; k++) {
        const double vi = iq_u8 ? iq_u8[2 * k] / 256.0 : iq_f64[2 * k];
        const double vq = iq_u8 ? iq_u8[2 * k + 1] / 256.0 : iq_f64[2 * k + 1];
        out[2 * k] = vi * c - vq * s + 0.5;
        out[2 * k + 1] = vi * s + vq * c + 0.5;
        const double ns = c * dsin + s * dcos;
        const double nc = c * dcos - s * dsin;
        s = ns;
        c = nc;
    }
    *cosine = c;
    *sine = s;
}

int orc_rows_shifted(const uint8_t *iq, size_t n_frames, int n, size_t hop, int flip, int mode,
                     double cycles_per_sample, double phase0_cycles, void *out) {
    return orc_rows_shifted_windowed(iq, n_frames, n, hop, flip, mode, cycles_per_sample, phase0_cycles, NULL, out);
}

int orc_rows_shifted_windowed(const uint8_t *iq, size_t n_frames, int n, size_t hop, int flip, int mode,
                              double cycles_per_sample, double phase0_cycles, const double *window, void *out) {
    orc_plan p;
    if (mode < 0 || mode > 5) return -2;
    if (orc_plan_init(&p, n) != 0) return -1;
    const long double tau = 6.283185307179586476925286766559L;
    double *shifted = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *tmp_in = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *tmp_out = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    size_t row_bytes = (mode == 1 || mode == 2) ? (size_t)n
                       : (mode == 3)            ? sizeof(double) * 2 * (size_t)n
                                                : sizeof(double) * (size_t)n;
    for (size_t f = 0; f < n_frames; f++) {
        for (int k = 0; k < n; k++) {
            const size_t m = f * hop + (size_t)k;
            const uint8_t bi = iq[2 * m], bq = iq[2 * m + 1];
            const double vi = (flip ? (uint8_t)(bi ^ 0x80) : bi) / 256.0;
            const double vq = (flip ? (uint8_t)(bq ^ 0x80) : bq) / 256.0;
            long double turns = (long double)phase0_cycles + (long double)cycles_per_sample * (long double)m;
            turns -= floorl(turns);
            const double c = (double)cosl(tau * turns), s = (double)sinl(tau * turns);
            shifted[2 * k] = vi * c - vq * s + 0.5;
            shifted[2 * k + 1] = vi * s + vq * c + 0.5;
        }
        orc_unpack_center_f64(shifted, (size_t)n, tmp_in); /* src/nrf.c:607-612 */
        if (window != NULL) {                              /* the weight beside the (-1)^ii */
            for (int k = 0; k < n; k++) {
                tmp_in[2 * k] *= window[k];
                tmp_in[2 * k + 1] *= window[k];
            }
        }
        orc_plan_exec(&p, tmp_in, tmp_out);
        orc_epilogue(n, mode, tmp_out, (uint8_t *)out + f * row_bytes);
    }
    free(shifted);
    free(tmp_in);
    free(tmp_out);
    orc_plan_free(&p);
    return 0;
}

/* ---- cpu_baseline ("port") ---------------------------------------------- */

static double now_seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double orc_time_mag_rows(const uint8_t *iq, size_t n_frames, int n, size_t hop,
                         double *sink) {
    orc_plan p;
    if (orc_plan_init(&p, n) != 0) return -1.0;
    uint8_t *tmp_u8 = (uint8_t *)malloc((size_t)2 * (size_t)n);
    double *tmp_in = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *tmp_out = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *row = (double *)malloc(sizeof(double) * (size_t)n);
    double t0 = now_seconds();
    for (size_t f = 0; f < n_frames; f++) {
        orc_one_row(&p, iq + 2 * f * hop, 1, 0, tmp_u8, tmp_in, tmp_out, row);
    }
    double t1 = now_seconds();
    if (sink) memcpy(sink, row, sizeof(double) * (size_t)n);
    free(tmp_u8);
    free(tmp_in);
    free(tmp_out);
    free(row);
    orc_plan_free(&p);
    return t1 - t0;
}

typedef struct {
    const uint8_t *iq;
    size_t f0, f1;
    int n;
    size_t hop;
    double checksum;
    pthread_barrier_t *gate; /* crossed twice: plans ready -> timed region starts; work done */
} orc_mt_job;

static void *orc_mt_worker(void *arg) {
    orc_mt_job *job = (orc_mt_job *)arg;
    orc_plan p;
    if (orc_plan_init(&p, job->n) != 0) {
        pthread_barrier_wait(job->gate);
        pthread_barrier_wait(job->gate);
        return NULL;
    }
    const int n = job->n;
    uint8_t *tmp_u8 = (uint8_t *)malloc((size_t)2 * (size_t)n);
    double *tmp_in = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *tmp_out = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *row = (double *)malloc(sizeof(double) * (size_t)n);
    double acc = 0;
    pthread_barrier_wait(job->gate);
    for (size_t f = job->f0; f < job->f1; f++) {
        orc_one_row(&p, job->iq + 2 * f * job->hop, 1, 0, tmp_u8, tmp_in, tmp_out, row);
        acc += row[1];
    }
    pthread_barrier_wait(job->gate);
    job->checksum = acc;
    free(tmp_u8);
    free(tmp_in);
    free(tmp_out);
    free(row);
    orc_plan_free(&p);
    return NULL;
}

/* Frames sharded over n_threads pthreads, one plan per thread (BASELINE.md
 * section 4, item 2 (ii)).  Returns wall seconds. */
double orc_time_mag_rows_mt(const uint8_t *iq, size_t n_frames, int n, size_t hop,
                            int n_threads, double *checksum) {
    if (n_threads < 1) n_threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    orc_mt_job *jobs = (orc_mt_job *)calloc((size_t)n_threads, sizeof(orc_mt_job));
    pthread_barrier_t gate;
    pthread_barrier_init(&gate, NULL, (unsigned)n_threads + 1u);
    for (int t = 0; t < n_threads; t++) {
        jobs[t].iq = iq;
        jobs[t].n = n;
        jobs[t].hop = hop;
        jobs[t].f0 = n_frames * (size_t)t / (size_t)n_threads;
        jobs[t].f1 = n_frames * (size_t)(t + 1) / (size_t)n_threads;
        jobs[t].gate = &gate;
        pthread_create(&th[t], NULL, orc_mt_worker, &jobs[t]);
    }
    /* planning (FFTW_MEASURE in the reference, src/nrf.c:564) is a one-off cost and stays outside */
    pthread_barrier_wait(&gate);
    double t0 = now_seconds();
    pthread_barrier_wait(&gate);
    double t1 = now_seconds();
    double acc = 0;
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        acc += jobs[t].checksum;
    }
    pthread_barrier_destroy(&gate);
    if (checksum) *checksum = acc;
    free(th);
    free(jobs);
    return t1 - t0;
}

/* ---- whole rows on several cores (the parity tests' full-size comparisons) ---- */

typedef struct {
    const uint8_t *iq;
    size_t f0, f1, hop, row_bytes;
    int n, flip, mode, rc;
    const double *window;
    uint8_t *out;
} orc_rows_job;

static void *orc_rows_worker(void *arg) {
    orc_rows_job *job = (orc_rows_job *)arg;
    /* the single-threaded function itself on this thread's frames: the rows cannot differ from orc_rows' */
    job->rc = orc_rows_windowed(job->iq + 2 * job->f0 * job->hop, job->f1 - job->f0, job->n, job->hop, job->flip,
                                job->mode, job->window, job->out + job->f0 * job->row_bytes);
    return NULL;
}

int orc_rows_mt(const uint8_t *iq, size_t n_frames, int n, size_t hop, int flip, int mode,
                const double *window, int n_threads, void *out) {
    if (mode < 0 || mode > 5) return -2;
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n_frames) n_threads = n_frames ? (int)n_frames : 1;
    const size_t row_bytes = (mode == 1 || mode == 2) ? (size_t)n
                             : (mode == 3)            ? sizeof(double) * 2 * (size_t)n
                                                      : sizeof(double) * (size_t)n;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    orc_rows_job *jobs = (orc_rows_job *)calloc((size_t)n_threads, sizeof(orc_rows_job));
    for (int t = 0; t < n_threads; t++) {
        jobs[t].iq = iq;
        jobs[t].f0 = n_frames * (size_t)t / (size_t)n_threads;
        jobs[t].f1 = n_frames * (size_t)(t + 1) / (size_t)n_threads;
        jobs[t].hop = hop;
        jobs[t].row_bytes = row_bytes;
        jobs[t].n = n;
        jobs[t].flip = flip;
        jobs[t].mode = mode;
        jobs[t].window = window;
        jobs[t].out = (uint8_t *)out;
        pthread_create(&th[t], NULL, orc_rows_worker, &jobs[t]);
    }
    int rc = 0;
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        if (jobs[t].rc != 0) rc = jobs[t].rc;
    }
    free(th);
    free(jobs);
    return rc;
}

/* ---- cpu_baseline through an FFTW3-API library (dlopen) ------------------- */

#include <dlfcn.h>

typedef void *(*fftw_plan_dft_1d_fn)(int, double *, double *, int, unsigned);
typedef void (*fftw_execute_fn)(void *);
typedef void (*fftw_destroy_plan_fn)(void *);

typedef struct {
    const uint8_t *iq;
    size_t f0, f1;
    int n;
    size_t hop;
    double checksum;
    double *rows;
    size_t rows_cap;
    int passes;
    fftw_plan_dft_1d_fn plan_fn;
    fftw_execute_fn exec_fn;
    fftw_destroy_plan_fn destroy_fn;
    pthread_mutex_t *planner;
    pthread_barrier_t *gate;
    int failed;
} orc_fftw_job;

static void *orc_fftw_worker(void *arg) {
    orc_fftw_job *job = (orc_fftw_job *)arg;
    const int n = job->n;
    uint8_t *tmp_u8 = (uint8_t *)malloc((size_t)2 * (size_t)n);
    double *in = NULL, *out = NULL;
    double *row = (double *)malloc(sizeof(double) * (size_t)n);
    void *plan = NULL;
    if (posix_memalign((void **)&in, 64, sizeof(double) * 2 * (size_t)n) == 0 &&
        posix_memalign((void **)&out, 64, sizeof(double) * 2 * (size_t)n) == 0) {
        pthread_mutex_lock(job->planner); /* FFTW's planner is not thread-safe */
        plan = job->plan_fn(n, in, out, -1 /* FFTW_FORWARD */, 0u /* FFTW_MEASURE */);
        pthread_mutex_unlock(job->planner);
    }
    if (plan == NULL) {
        job->failed = 1;
        pthread_barrier_wait(job->gate);
        pthread_barrier_wait(job->gate);
        return NULL;
    }
    double acc = 0;
    pthread_barrier_wait(job->gate);
    for (int pass = 0; pass < job->passes; pass++) {
        for (size_t f = job->f0; f < job->f1; f++) {
            orc_flip_u8(job->iq + 2 * f * job->hop, tmp_u8, (size_t)2 * (size_t)n); /* a1 */
            orc_unpack_center_u8(tmp_u8, (size_t)n, in);                             /* a4 */
            job->exec_fn(plan);                                                      /* a5 */
            orc_mag_row(out, n, row);                                                /* a7 */
            acc += row[1];
            if (job->rows != NULL && f < job->rows_cap) {
                memcpy(job->rows + f * (size_t)n, row, sizeof(double) * (size_t)n);
            }
        }
    }
    pthread_barrier_wait(job->gate);
    job->checksum = acc;
    pthread_mutex_lock(job->planner);
    job->destroy_fn(plan);
    pthread_mutex_unlock(job->planner);
    free(tmp_u8);
    free(in);
    free(out);
    free(row);
    return NULL;
}

double orc_time_mag_rows_fftw(const char *lib, const uint8_t *iq, size_t n_frames, int n, size_t hop,
                              int n_threads, int passes, double *rows, size_t rows_cap, double *checksum) {
    void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
    if (h == NULL) return -1.0;
    fftw_plan_dft_1d_fn plan_fn;
    fftw_execute_fn exec_fn;
    fftw_destroy_plan_fn destroy_fn;
    /* the POSIX idiom for dlsym -> function pointer under -pedantic */
    *(void **)(&plan_fn) = dlsym(h, "fftw_plan_dft_1d");
    *(void **)(&exec_fn) = dlsym(h, "fftw_execute");
    *(void **)(&destroy_fn) = dlsym(h, "fftw_destroy_plan");
    if (plan_fn == NULL || exec_fn == NULL || destroy_fn == NULL) return -2.0;
    if (n_threads < 1) n_threads = 1;
    pthread_mutex_t planner;
    pthread_mutex_init(&planner, NULL);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    orc_fftw_job *jobs = (orc_fftw_job *)calloc((size_t)n_threads, sizeof(orc_fftw_job));
    pthread_barrier_t gate;
    pthread_barrier_init(&gate, NULL, (unsigned)n_threads + 1u);
    for (int t = 0; t < n_threads; t++) {
        jobs[t].iq = iq;
        jobs[t].n = n;
        jobs[t].hop = hop;
        jobs[t].f0 = n_frames * (size_t)t / (size_t)n_threads;
        jobs[t].f1 = n_frames * (size_t)(t + 1) / (size_t)n_threads;
        jobs[t].rows = rows;
        jobs[t].rows_cap = rows_cap;
        jobs[t].passes = passes < 1 ? 1 : passes;
        jobs[t].plan_fn = plan_fn;
        jobs[t].exec_fn = exec_fn;
        jobs[t].destroy_fn = destroy_fn;
        jobs[t].planner = &planner;
        jobs[t].gate = &gate;
        pthread_create(&th[t], NULL, orc_fftw_worker, &jobs[t]);
    }
    pthread_barrier_wait(&gate); /* every thread has its plan: planning stays outside the timed region */
    double t0 = now_seconds();
    pthread_barrier_wait(&gate);
    double t1 = now_seconds();
    double acc = 0;
    int failed = 0;
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        acc += jobs[t].checksum;
        failed |= jobs[t].failed;
    }
    pthread_barrier_destroy(&gate);
    if (checksum) *checksum = acc;
    free(th);
    free(jobs);
    pthread_mutex_destroy(&planner);
    return failed ? -3.0 : t1 - t0;
}
