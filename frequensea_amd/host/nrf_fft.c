/*
 * nrf_fft.c -- frequensea's nrf_fft block with the spectrum computed by
 * libfsea_hip.so on an MI355X (include/nrf.h, include/fsea.h).
 *
 * Reference behaviour restated (paths under /root/reference):
 *   nrf_fft_new         src/nrf.c:557-567
 *   nrf_fft_shift       src/nrf.c:569-596
 *   nrf_fft_process     src/nrf.c:598-631
 *   nrf_fft_get_buffer  src/nrf.c:633-635
 *   nrf_fft_free        src/nrf.c:637-642 (which leaks `buffer`; fixed here)
 * What moved to the GPU: the unpack/centre loop, the FFT and the magnitude +
 * DC patch (one fused kernel).  What changed on the host: the history is a
 * ring (the reference memmoves the whole history, 8 MiB at 1024x1024, per
 * row: src/nrf.c:617), linearised only in nrf_fft_get_buffer.
 *
 * NRF_FFT_HISTORY=device in the environment keeps that ring in HBM instead
 * (fsea_history_*, include/fsea.h): process writes the row in place on the
 * device, shift is a kernel, get_buffer is one device-to-host transfer of H*N
 * f32 plus one widening.  Same results bit for bit; which one is faster
 * depends on how often get_buffer is called per process (profiles/,
 * scripts/nrf_latency.py): the default is the host ring.
 *
 * NRF_FFT_WINDOW=hann|hamming|blackman|blackmanharris|flattop in the environment
 * puts that taper beside the (-1)^ii of the unpack loop (src/nrf.c:601-614),
 * for U8 and F64 buffers alike (so the nrf_freq_shifter -> nrf_fft chain of
 * lua/fft-shifted.lua:52-55 is tapered too) and in both history modes:
 * fsea_plan_set_window on the block's plan, fused into the kernel's pass 0.
 * Unset (or "rect" / "none") is the reference: it has no taper, the API keeps
 * its signatures, and the scenes need no change to get one.  That variable is
 * only the DEFAULT of new blocks: nrf_fft_set_window(fft, name) and
 * nrf_fft_set_window_weights(fft, w) -- two ADDITIONS beside the reference's five
 * prototypes (src/nrf.h:138-142), nothing existing changes -- choose the taper
 * per nrf_fft object, at any time between process calls, in both history modes.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fsea.h"
#include "nrf.h"
#ifdef FSEA_NRF_FFT_ONLY
/* libfsea_nrf_fft.so: only the five nrf_fft_* functions, to be linked next to the application's own
 * nut.c / nrf.c; then only the public nut.h interface is available. */
#define nut_private_new_f64_unfilled(n_elements, n_channels) nut_buffer_new_f64((n_elements), (n_channels), NULL)
#else
#include "nut_private.h"
#endif

static void fsea_fatal(const char *what, int rc) {
    /* same convention as src/nrf.c:54-78: print and exit */
    fprintf(stderr, "NRF FFT fatal error: %s failed (%d): %s\n", what, rc, fsea_last_error_string());
    exit(EXIT_FAILURE);
}

/* taper names of nrf_fft_set_window / NRF_FFT_WINDOW -> fsea_window_fill kinds; -1 = rectangular (no taper), -2 = unknown */
static int fsea_window_kind(const char *name) {
    static const struct { const char *name; int kind; } tapers[] = {
        {"hann", FSEA_WINDOW_HANN},           {"hamming", FSEA_WINDOW_HAMMING}, {"blackman", FSEA_WINDOW_BLACKMAN},
        {"blackmanharris", FSEA_WINDOW_BLACKMANHARRIS}, {"flattop", FSEA_WINDOW_FLATTOP}};
    if (name == NULL || name[0] == '\0' || strcmp(name, "rect") == 0 || strcmp(name, "none") == 0) return -1;
    for (size_t i = 0; i < sizeof(tapers) / sizeof(tapers[0]); i++) {
        if (strcmp(name, tapers[i].name) == 0) return tapers[i].kind;
    }
    return -2;
}

static void set_window_by_name(fsea_plan *plan, int fft_size, const char *name, const char *who) {
    const int kind = fsea_window_kind(name);
    int rc;
    if (kind == -1) {
        rc = fsea_plan_set_window(plan, NULL);
        if (rc != FSEA_OK) fsea_fatal("fsea_plan_set_window", rc);
        return;
    }
    float *w = (float *)malloc(sizeof(float) * (size_t)fft_size);
    if (w == NULL) {
        fprintf(stderr, "NRF FFT fatal error: out of memory\n");
        exit(EXIT_FAILURE);
    }
    rc = fsea_window_fill(kind, fft_size, w);
    if (rc != FSEA_OK) fsea_fatal("fsea_window_fill", rc);
    rc = fsea_plan_set_window(plan, w); /* fails for a size without a kernel of its own: said loudly, not ignored */
    if (rc != FSEA_OK) {
        char what[96];
        snprintf(what, sizeof(what), "fsea_plan_set_window (%s)", who);
        fsea_fatal(what, rc);
    }
    free(w);
}

void nrf_fft_set_window(nrf_fft *fft, const char *name) {
    if (fsea_window_kind(name) == -2) {
        /* a programming error in the scene, handled as src/main.cpp handles a wrong argument type: say it and stop */
        fprintf(stderr, "NRF FFT fatal error: nrf_fft_set_window: \"%s\" is not one of hann, hamming, blackman, blackmanharris, "
                        "flattop, rect\n", name);
        exit(EXIT_FAILURE);
    }
    /* under the block's mutex: a device thread may be inside nrf_fft_process (SURVEY 8(b) threading); rows already in the
     * history stay as they were computed, the next process call uses the new taper */
    pthread_mutex_lock(&fft->mutex);
    set_window_by_name((fsea_plan *)fft->backend, fft->fft_size, name, "nrf_fft_set_window");
    pthread_mutex_unlock(&fft->mutex);
}

void nrf_fft_set_window_weights(nrf_fft *fft, const float *weights) {
    pthread_mutex_lock(&fft->mutex);
    const int rc = fsea_plan_set_window((fsea_plan *)fft->backend, weights); /* NULL: back to the reference's rectangular frames */
    if (rc != FSEA_OK) fsea_fatal("fsea_plan_set_window (nrf_fft_set_window_weights)", rc);
    pthread_mutex_unlock(&fft->mutex);
}

nrf_fft *nrf_fft_new(int fft_size, int fft_history_size) {
    nrf_fft *fft = (nrf_fft *)calloc(1, sizeof(nrf_fft));
    if (fft == NULL) {
        fprintf(stderr, "NRF FFT fatal error: out of memory\n");
        exit(EXIT_FAILURE);
    }
    nrf_block_init(&fft->block, NRF_BLOCK_GENERIC, (nrf_block_process_fn)nrf_fft_process,
                   (nrf_block_result_fn)nrf_fft_get_buffer);
    fft->fft_size = fft_size;
    fft->fft_history_size = fft_history_size;
    fsea_plan *plan = NULL;
    const char *dev_env = getenv("NRF_FFT_DEVICE"); /* which GPU; the reference has no such notion */
    int rc = fsea_plan_create(&plan, fft_size, fft_size, FSEA_MODE_MAG_F32, dev_env ? atoi(dev_env) : 0);
    if (rc != FSEA_OK) fsea_fatal("fsea_plan_create", rc);
    fft->backend = plan;
    /* the process-wide default for scenes that do not ask (an unmodified fft-sea.lua gets a taper this way); a scene or a
     * C caller that wants its own calls nrf_fft_set_window afterwards */
    const char *win_env = getenv("NRF_FFT_WINDOW");
    if (win_env != NULL && win_env[0] != '\0') {
        if (fsea_window_kind(win_env) == -2) {
            fprintf(stderr, "NRF FFT fatal error: NRF_FFT_WINDOW=%s is not one of hann, hamming, blackman, blackmanharris, "
                            "flattop, rect\n", win_env);
            exit(EXIT_FAILURE);
        }
        set_window_by_name(plan, fft_size, win_env, "NRF_FFT_WINDOW");
    }
    const char *hist_env = getenv("NRF_FFT_HISTORY");
    if (hist_env != NULL && strcmp(hist_env, "device") == 0) {
        fsea_history *hist = NULL;
        rc = fsea_history_create(plan, fft_history_size, &hist);
        if (rc != FSEA_OK) fsea_fatal("fsea_history_create", rc);
        fft->device_history = hist;
    }
    fft->buffer = fft->device_history != NULL
                      ? NULL
                      : (double *)calloc((size_t)fft_size * (size_t)fft_history_size, sizeof(double));
    fft->row_f32 = (float *)calloc((size_t)fft_size, sizeof(float));
    fft->scratch = calloc((size_t)fft_size * 2, sizeof(double));
    if ((fft->buffer == NULL && fft->device_history == NULL) || fft->row_f32 == NULL || fft->scratch == NULL) {
        fprintf(stderr, "NRF FFT fatal error: out of memory\n");
        exit(EXIT_FAILURE);
    }
    fft->ring_head = 0;
    pthread_mutex_init(&fft->mutex, NULL);
    return fft;
}

void nrf_fft_shift(nrf_fft *fft, double d) {
    const int n = fft->fft_size;
    const int shift = (int)round(n / d);
    if (shift == 0) return;
    pthread_mutex_lock(&fft->mutex);
    if (fft->device_history != NULL) {
        const int rc = fsea_history_shift((fsea_history *)fft->device_history, shift);
        if (rc != FSEA_OK) fsea_fatal("fsea_history_shift", rc);
    } else if (abs(shift) >= n) {
        /* shifted out of range: start over */
        memset(fft->buffer, 0, sizeof(double) * (size_t)n * (size_t)fft->fft_history_size);
    } else {
        /* the per-row operation does not depend on the row order, so it is
         * applied to the ring as stored */
        for (int y = 0; y < fft->fft_history_size; y++) {
            double *row = fft->buffer + (size_t)y * (size_t)n;
            if (shift > 0) {
                memmove(row, row + shift, sizeof(double) * (size_t)(n - shift));
                memset(row + (n - shift), 0, sizeof(double) * (size_t)shift);
            } else {
                memmove(row - shift, row, sizeof(double) * (size_t)(n + shift));
                memset(row, 0, sizeof(double) * (size_t)(-shift));
            }
        }
    }
    pthread_mutex_unlock(&fft->mutex);
}

void nrf_fft_process(nrf_fft *fft, nut_buffer *buffer) {
    const int n = fft->fft_size;
    fsea_plan *plan = (fsea_plan *)fft->backend;
    /* The reference unpacks every sample of the buffer but transforms only the
     * first fft_size (src/nrf.c:599-615); a shorter buffer is zero-padded here
     * (the reference would transform stale samples of an earlier call). */
    const int have = (buffer->length * buffer->channels) / 2;
    pthread_mutex_lock(&fft->mutex);
    int rc;
    if (buffer->type == NUT_BUFFER_U8) {
        const uint8_t *iq = buffer->data.u8;
        if (have < n) {
            /* missing samples are 0.0, i.e. u8 value 0 under x = u8 / 256.0 */
            uint8_t *pad = (uint8_t *)fft->scratch;
            memset(pad, 0, (size_t)n * 2);
            memcpy(pad, buffer->data.u8, (size_t)have * 2);
            iq = pad;
        }
        /* device buffers are already offset binary (src/nrf.c:103-106) -> flip = 0 */
        if (fft->device_history != NULL) {
            rc = fsea_history_push_u8_host((fsea_history *)fft->device_history, iq, 0);
        } else {
            rc = fsea_exec_u8_host(plan, iq, 1, 0, fft->row_f32);
        }
        if (rc != FSEA_OK) fsea_fatal("fsea_exec_u8_host", rc);
    } else {
        const double *iq = buffer->data.f64;
        if (have < n) {
            double *pad = (double *)fft->scratch;
            memset(pad, 0, sizeof(double) * (size_t)n * 2);
            memcpy(pad, buffer->data.f64, sizeof(double) * (size_t)have * 2);
            iq = pad;
        }
        if (fft->device_history != NULL) {
            rc = fsea_history_push_f64_host((fsea_history *)fft->device_history, iq);
        } else {
            rc = fsea_exec_f64_host(plan, iq, 1, fft->row_f32);
        }
        if (rc != FSEA_OK) fsea_fatal("fsea_exec_f64_host", rc);
    }
    if (fft->device_history == NULL) {
        /* push as the newest row: the ring head moves back by one */
        fft->ring_head = (fft->ring_head + fft->fft_history_size - 1) % fft->fft_history_size;
        double *row = fft->buffer + (size_t)fft->ring_head * (size_t)n;
        for (int i = 0; i < n; i++) row[i] = (double)fft->row_f32[i];
    }
    pthread_mutex_unlock(&fft->mutex);
}

nut_buffer *nrf_fft_get_buffer(nrf_fft *fft) {
    const int n = fft->fft_size, h = fft->fft_history_size;
    nut_buffer *out = nut_private_new_f64_unfilled(n * h, 1); /* both memcpys below cover it entirely */
    pthread_mutex_lock(&fft->mutex);
    if (fft->device_history != NULL) {
        const int rc = fsea_history_get_f64((fsea_history *)fft->device_history, out->data.f64);
        if (rc != FSEA_OK) fsea_fatal("fsea_history_get_f64", rc);
        pthread_mutex_unlock(&fft->mutex);
        return out;
    }
    const int first = h - fft->ring_head; /* rows from the head to the end of storage */
    memcpy(out->data.f64, fft->buffer + (size_t)fft->ring_head * (size_t)n, sizeof(double) * (size_t)first * (size_t)n);
    memcpy(out->data.f64 + (size_t)first * (size_t)n, fft->buffer, sizeof(double) * (size_t)fft->ring_head * (size_t)n);
    pthread_mutex_unlock(&fft->mutex);
    return out;
}

void nrf_fft_free(nrf_fft *fft) {
    if (fft == NULL) return;
    fsea_history_destroy((fsea_history *)fft->device_history);
    fsea_plan_destroy((fsea_plan *)fft->backend);
    pthread_mutex_destroy(&fft->mutex);
    free(fft->buffer);
    free(fft->row_f32);
    free(fft->scratch);
    free(fft);
}
