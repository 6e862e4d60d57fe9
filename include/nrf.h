/*
 * nrf.h -- the part of frequensea's SDR block library that sits on the
 * IQ-FFT path, with the spectrum computed on an MI355X instead of by FFTW.
 *
 * Interface being replaced (paths under /root/reference):
 *   src/nrf.h:19-23    NRF_* constants
 *   src/nrf.h:25-52    nrf_block + NRF_BLOCK (must stay the FIRST member of
 *                      every block struct: src/main.cpp:616-624 reads
 *                      block->type through the Lua table's __ptr__)
 *   src/nrf.h:56-103   nrf_device (file-replay "dummy" source only; the
 *                      HackRF / RTL-SDR drivers are out of scope)
 *   src/nrf.h:128-142  nrf_fft: nrf_fft_new / _shift / _process /
 *                      _get_buffer / _free -- identical signatures
 *   src/nrf.h:192-207  nrf_freq_shifter, the block lua/fft-shifted.lua puts in
 *                      front of nrf_fft (host arithmetic, as in the reference)
 * Differences, all invisible to callers: <fftw3.h> is gone, the FFTW-typed
 * members of nrf_fft (touched by nobody outside src/nrf.c) became an opaque
 * backend handle, the history is a ring instead of an 8 MiB memmove per row,
 * and a mutex serialises process / get_buffer / shift (the reference has a
 * latent race there: src/nrf.c:37-50 vs src/main.cpp:801).
 *
 * Error convention as in the reference (src/nrf.c:54-78): these functions do
 * not return status codes; a fatal backend error (no GPU, HIP failure) prints
 * to stderr and exit(EXIT_FAILURE)s.  There is no CPU fallback.
 */
#ifndef NRF_H
#define NRF_H

#include <pthread.h>
#include <stdint.h>

#include "nut.h"

#define NRF_BUFFER_SIZE_BYTES (16 * 16384) /* one device block: 262144 bytes */
#define NRF_SAMPLES_LENGTH 131072          /* IQ samples per block */
#define NRF_IQ_RESOLUTION 256
#define DEFAULT_FFT_SIZE 128
#define DEFAULT_FFT_HISTORY_SIZE 128

/* ---- block graph (src/nrf.h:25-52, src/nrf.c:24-50) ------------------- */

#define NRF_BLOCK_MAX_OUTPUTS 10

typedef enum {
    NRF_BLOCK_SOURCE = 1,
    NRF_BLOCK_GENERIC,
    NRF_BLOCK_SINK
} nrf_block_type;

typedef struct nrf_block nrf_block;
typedef void (*nrf_block_process_fn)(nrf_block *block, nut_buffer *buffer);
typedef nut_buffer *(*nrf_block_result_fn)(void *block);

struct nrf_block {
    nrf_block_type type;
    nrf_block_process_fn process_fn;
    nrf_block_result_fn result_fn;
    int n_outputs;
    void *outputs[NRF_BLOCK_MAX_OUTPUTS];
};

void nrf_block_init(nrf_block *block, nrf_block_type type, nrf_block_process_fn process_fn,
                    nrf_block_result_fn result_fn);
void nrf_block_connect(nrf_block *input, nrf_block *output);
/* process_fn(block, buffer); then, if the block has outputs, push result_fn's
 * buffer to each of them and free it. */
void nrf_block_process(nrf_block *block, nut_buffer *buffer);

#define NRF_BLOCK nrf_block block

/* ---- sample source (src/nrf.h:56-103) ---------------------------------- */

typedef struct {
    int sample_rate;
    double freq_mhz;
    const char *data_file;
} nrf_device_config;

typedef enum {
    NRF_DEVICE_DUMMY = 0,
    NRF_DEVICE_RTLSDR,
    NRF_DEVICE_HACKRF
} nrf_device_type;

typedef struct nrf_device nrf_device;
typedef void (*nrf_device_decode_cb_fn)(nrf_device *device, void *ctx);

struct nrf_device {
    NRF_BLOCK;
    nrf_device_type device_type; /* always NRF_DEVICE_DUMMY in this build */
    void *device;
    int sample_rate;

    nrf_device_decode_cb_fn decode_cb_fn;
    void *decode_cb_ctx;

    pthread_t receive_thread;
    pthread_mutex_t data_mutex;
    int receiving;
    int paused;

    uint8_t *receive_buffer; /* whole replay file, raw HackRF int8 bytes */
    int dummy_block_length;  /* blocks in receive_buffer */
    int dummy_block_index;

    uint8_t samples[NRF_BUFFER_SIZE_BYTES]; /* current block, offset binary */
};

/* Replays `data_file` (raw int8 IQ as written by c/rfcap.c) in 262144-byte
 * blocks at 60 Hz on its own thread, flipping each byte to offset binary
 * (src/nrf.c:95-110, 162-170, 256-284).  A missing file gives one zero block. */
nrf_device *nrf_device_new(double freq_mhz, const char *data_file);
nrf_device *nrf_device_new_with_config(nrf_device_config config);
double nrf_device_set_frequency(nrf_device *device, double freq_mhz);
void nrf_device_set_decode_handler(nrf_device *device, nrf_device_decode_cb_fn fn, void *ctx);
void nrf_device_set_paused(nrf_device *device, int paused);
void nrf_device_step(nrf_device *device);
/* Locked snapshot: u8, length NRF_SAMPLES_LENGTH, 2 channels (src/nrf.c:352-357). */
nut_buffer *nrf_device_get_samples_buffer(nrf_device *device);
void nrf_device_free(nrf_device *device);

/* ---- FFT analysis (src/nrf.h:128-142, src/nrf.c:557-642) ---------------- */

typedef struct {
    NRF_BLOCK;
    int fft_size;
    int fft_history_size;
    double *buffer;        /* history ring: fft_history_size rows of fft_size */
    int ring_head;         /* ring row that holds the newest spectrum */
    void *backend;         /* fsea_plan* (libfsea_hip.so) */
    float *row_f32;        /* staging for one device row */
    void *scratch;         /* zero-padded / converted input for short buffers */
    pthread_mutex_t mutex;
    void *device_history;  /* fsea_history* when NRF_FFT_HISTORY=device: the ring lives in HBM, `buffer` is NULL */
} nrf_fft;

/* Plan + zeroed history of fft_history_size rows.  Exits if no GPU. */
nrf_fft *nrf_fft_new(int fft_size, int fft_history_size);
/* Scroll every history row by round(fft_size / d) bins (src/nrf.c:569-596). */
void nrf_fft_shift(nrf_fft *fft, double d);
/* One new row from the first fft_size samples of `buffer` (u8 IQ as produced by
 * nrf_device_get_samples_buffer, or f64 IQ): x[n] = (-1)^n * u8/256, forward
 * DFT, magnitude, bin N/2 := bin N/2-1; pushed as row 0 (src/nrf.c:598-631). */
void nrf_fft_process(nrf_fft *fft, nut_buffer *buffer);
/* Fresh f64 copy of the whole history, newest row first; caller frees with
 * nut_buffer_free (src/nrf.c:633-635). */
nut_buffer *nrf_fft_get_buffer(nrf_fft *fft);
void nrf_fft_free(nrf_fft *fft);
/* ADDITIONS (not in the reference; the five prototypes above are src/nrf.h:138-142 unchanged).  BASELINE's north_star
 * names a "windowed 1D FFT"; the reference's only pre-FFT weight is powf(-1, ii) (src/nrf.c:611-612), i.e. rectangular.
 * A taper w[n] rides beside that sign in the kernel's fused unpack prologue: x[n] = (-1)^n w[n] u8[n]/256 (fsea.h:
 * fsea_plan_set_window).  Per nrf_fft object; callable at any time, also between nrf_fft_process calls and from another
 * thread (taken under the block's mutex): rows already in the history keep the taper they were computed with.
 *   name: "hann", "hamming", "blackman", "blackmanharris", "flattop" (periodic, scipy.signal.get_window's values), or
 *         "rect" / "none" / "" / NULL = the reference's rectangular frames.  Anything else: message on stderr + exit, the
 *         reference's convention for a wrong argument (src/main.cpp:46-60).
 *   weights: fft_size floats, copied; NULL = rectangular.
 * The environment variable NRF_FFT_WINDOW=<name> is the default every nrf_fft_new starts from (unmodified scenes). */
void nrf_fft_set_window(nrf_fft *fft, const char *name);
void nrf_fft_set_window_weights(nrf_fft *fft, const float *weights);

/* ---- frequency shifter (src/nrf.h:192-207, src/nrf.c:817-870) ------------ */

typedef struct {
    NRF_BLOCK;
    int freq_offset; /* Hz */
    int sample_rate; /* Hz */
    double cosine;   /* phase carried from block to block, starts at (1, 0) */
    double sine;
    nut_buffer *buffer; /* last output, F64 */
} nrf_freq_shifter;

nrf_freq_shifter *nrf_freq_shifter_new(int freq_offset, int sample_rate);
/* In place on separate I and Q arrays: (i, q) <- (i, q) rotated by the running phase, which
 * advances by 2 pi freq_offset / sample_rate per sample.  No offset is added. */
void nrf_freq_shifter_process_samples(nrf_freq_shifter *shifter, double *samples_i, double *samples_q, int length);
/* Interleaved 2-channel buffer (u8 values count as u8 / 256.0): rotated samples + 0.5 on both
 * components go to the shifter's own F64 buffer.  As in the reference that buffer is created with
 * length = buffer->length * 2 and 2 channels, i.e. twice the room it needs; only the first half is
 * written (src/nrf.c:851).  nrf_fft_process reads just the first fft_size samples of it. */
void nrf_freq_shifter_process(nrf_freq_shifter *shifter, nut_buffer *buffer);
/* Copy of the last output (NULL before the first process call). */
nut_buffer *nrf_freq_shifter_get_buffer(nrf_freq_shifter *shifter);
void nrf_freq_shifter_free(nrf_freq_shifter *shifter);

#endif /* NRF_H */
