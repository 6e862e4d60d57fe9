/*
 * fsea-fft-batch -- batch export of FFT data as images, from capture files, on an MI355X.
 *
 * Re-statement of the reference's sweep tools with the HackRF replaced by recorded captures
 * (a capture = consecutive 262144-byte HackRF transfers, the format c/rfcap.c writes):
 *   default   c/fft-batch.c        1024-pt, 16384 rows, 10*log10(p+1e-20)*10, "fft-%.4f.png"
 *   --broad   c/fft-batch-broad.c   256-pt,  4096 rows, ...*5, pixel N/2 := pixel N/2-1,
 *                                   100-row gate (mean |X| < 1.1 -> skip), "broad-%.0f.png"
 * As in the reference one spectrum row is made from the first FFT_SIZE samples of each
 * transfer (c/fft-batch.c:62-69), the first 10 transfers after a retune are skipped
 * (:56-59, SAMPLE_BLOCKS_TO_SKIP) and the newest row is image row 0 (:72-74).
 * The per-sample loops and FFTW are one fused GPU launch per centre frequency (include/fsea.h), and the three host
 * stages overlap (pipeline.h): capture k + 1 is being read and the PNG of capture k - 1 encoded while k is on the GPU.
 * Consecutive captures go to the GPU ALTERNATELY ON TWO STREAMS with double-buffered device memory (round 4): upload,
 * gate, transform and download of capture k + 1 are queued while capture k is still draining -- the streaming shape of
 * the reference's receive callback (c/fft-batch.c:54-102: a transfer in, a row out) at the granularity of a capture.
 *
 * usage: fsea-fft-batch [--broad] [--rows H] [--fft N] [--skip K] [--out DIR] [--device D] [--timing] [--window NAME]
 *                       FREQ_MHZ=capture.raw [FREQ_MHZ=capture.raw ...]
 *   --window  hann | hamming | blackman | blackmanharris | flattop: a taper beside the reference's (-1)^n, fused into the
 *             kernel (fsea_plan_set_window; the reference's tools are rectangular, c/fft-batch.c:65-66 -- the default)
 *   --timing  print, at the end, the seconds each of the three stages was busy and the wall time of the loop
 */
#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include "easypng.h"
#include "fsea.h"
#include "pipeline.h"

#define TRANSFER_BYTES 262144 /* one HackRF transfer: 131072 IQ samples */
#define EVALUATE_ROWS 100     /* c/fft-batch-broad.c:22 */

static void die(const char *what) {
    fprintf(stderr, "fsea-fft-batch: %s: %s\n", what, fsea_last_error_string());
    exit(EXIT_FAILURE);
}

/* --window NAME: the periodic cosine-sum taper of that name on the plan (include/fsea.h: fsea_window_fill) */
static int set_named_window(fsea_plan *plan, const char *name, int n) {
    static const char *names[] = {"rect", "hann", "hamming", "blackman", "blackmanharris", "flattop"};
    for (int k = 0; k < 6; k++) {
        if (strcmp(name, names[k]) != 0) continue;
        float *w = (float *)malloc(sizeof(float) * (size_t)n);
        if (!w) return -1;
        int rc = fsea_window_fill(k, n, w);
        if (rc == 0 && k != 0) rc = fsea_plan_set_window(plan, w);
        free(w);
        return rc;
    }
    fprintf(stderr, "fsea-fft-batch: unknown window '%s' (hann, hamming, blackman, blackmanharris, flattop)\n", name);
    return -1;
}

typedef struct {
    char **args;          /* FREQ_MHZ=capture.raw, one per item */
    int rows_wanted, skip;
    size_t row_in;
} batch_ctx;

/* reader thread: rows of one capture, newest first: row y <- first 2N bytes of transfer skip + rows - 1 - y
 * (c/fft-batch.c:62-74).  0 = loaded, > 0 = fatal (cannot open, short read), < 0 = too few transfers (skip it). */
static int load_capture(void *vctx, int item, uint8_t *packed, int *rows_out) {
    const batch_ctx *ctx = (const batch_ctx *)vctx;
    const char *path = strchr(ctx->args[item], '=') + 1;
    /* one pread per row (a row is the first 2N bytes of a 262144-byte transfer): stdio's fseek + fread pair refills its
     * buffer for every row, twice the system calls for the 4.9 million rows of the reference's narrow sweep */
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        fprintf(stderr, "fsea-fft-batch: cannot open %s\n", path);
        return 1;
    }
    struct stat st;
    if (fstat(fd, &st) != 0) {
        fprintf(stderr, "fsea-fft-batch: cannot stat %s\n", path);
        close(fd);
        return 1;
    }
    const long transfers = (long)(st.st_size / TRANSFER_BYTES);
    int rows = (int)(transfers - ctx->skip);
    if (rows > ctx->rows_wanted) rows = ctx->rows_wanted;
    if (rows <= 0) {
        fprintf(stderr, "fsea-fft-batch: %s holds %ld transfers, need more than %d\n", path, transfers, ctx->skip);
        close(fd);
        return -1;
    }
    for (int y = 0; y < rows; y++) {
        const off_t tr = (off_t)ctx->skip + rows - 1 - y;
        uint8_t *dst = packed + (size_t)y * ctx->row_in;
        size_t got = 0;
        while (got < ctx->row_in) {
            const ssize_t r = pread(fd, dst + got, ctx->row_in - got, tr * (off_t)TRANSFER_BYTES + (off_t)got);
            if (r <= 0) break;
            got += (size_t)r;
        }
        if (got != ctx->row_in) {
            fprintf(stderr, "Short read, samples lost, exiting!\n");
            close(fd);
            return 1;
        }
    }
    close(fd);
    *rows_out = rows;
    return 0;
}

int main(int argc, char **argv) {
    int broad = 0, rows_wanted = -1, fft_size = -1, skip = 10, device = 0, timing = 0;
    const char *out_dir = ".", *window = NULL;
    int first_capture = argc;
    for (int i = 1; i < argc; i++) {
        if (strcmp(argv[i], "--broad") == 0) broad = 1;
        else if (strcmp(argv[i], "--rows") == 0 && i + 1 < argc) rows_wanted = atoi(argv[++i]);
        else if (strcmp(argv[i], "--fft") == 0 && i + 1 < argc) fft_size = atoi(argv[++i]);
        else if (strcmp(argv[i], "--skip") == 0 && i + 1 < argc) skip = atoi(argv[++i]);
        else if (strcmp(argv[i], "--out") == 0 && i + 1 < argc) out_dir = argv[++i];
        else if (strcmp(argv[i], "--device") == 0 && i + 1 < argc) device = atoi(argv[++i]);
        else if (strcmp(argv[i], "--timing") == 0) timing = 1;
        else if (strcmp(argv[i], "--window") == 0 && i + 1 < argc) window = argv[++i];
        else { first_capture = i; break; }
    }
    if (first_capture >= argc) {
        fprintf(stderr, "usage: fsea-fft-batch [--broad] [--rows H] [--fft N] [--skip K] [--out DIR] "
                        "[--device D] [--timing] [--window NAME] FREQ_MHZ=capture.raw ...\n");
        return EXIT_FAILURE;
    }
    if (fft_size < 0) fft_size = broad ? 256 : 1024;          /* FFT_SIZE */
    if (rows_wanted < 0) rows_wanted = broad ? 4096 : 16384;   /* FFT_HISTORY_SIZE */
    const size_t row_in = (size_t)2 * (size_t)fft_size;

    fsea_plan *plan = NULL;
    if (fsea_plan_create(&plan, fft_size, fft_size, broad ? FSEA_MODE_DB5_U8_DCFIX : FSEA_MODE_DB10_U8, device) != 0) {
        die("fsea_plan_create");
    }
    if (window && set_named_window(plan, window, fft_size) != 0) die("--window");
    /* two GPU slots: stream + device buffers each; capture k uses slot (number of captures sent to the GPU so far) % 2 */
    typedef struct {
        void *stream, *d_iq, *d_px;
        int busy, rows, job;      /* busy: a transform + download is queued on `stream`, to become PNG job `job` */
        char file_name[512];
    } gpu_slot;
    gpu_slot slot[2];
    memset(slot, 0, sizeof(slot));
    for (int k = 0; k < 2; k++) {
        if (fsea_stream_create(device, &slot[k].stream) != 0) die("fsea_stream_create");
        if (fsea_device_alloc(device, (size_t)rows_wanted * row_in, &slot[k].d_iq) != 0) die("fsea_device_alloc");
        if (fsea_device_alloc(device, (size_t)rows_wanted * (size_t)fft_size, &slot[k].d_px) != 0) die("fsea_device_alloc");
    }
    void *packed[2] = {NULL, NULL}, *pixels[2] = {NULL, NULL};
    for (int k = 0; k < 2; k++) {
        if (fsea_host_alloc((size_t)rows_wanted * row_in, &packed[k]) != 0) die("fsea_host_alloc");
        if (fsea_host_alloc((size_t)rows_wanted * (size_t)fft_size, &pixels[k]) != 0) die("fsea_host_alloc");
    }
    for (int i = first_capture; i < argc; i++) {
        if (!strchr(argv[i], '=')) {
            fprintf(stderr, "fsea-fft-batch: expected FREQ_MHZ=capture.raw, got %s\n", argv[i]);
            return EXIT_FAILURE;
        }
    }
    batch_ctx ctx = {argv + first_capture, rows_wanted, skip, row_in};
    capture_reader reader;
    png_writer writer;
    if (capture_reader_start(&reader, argc - first_capture, load_capture, &ctx, (uint8_t *)packed[0], (uint8_t *)packed[1]) != 0 ||
        png_writer_start(&writer, (uint8_t *)pixels[0], (uint8_t *)pixels[1]) != 0) {
        fprintf(stderr, "fsea-fft-batch: cannot start the reader / writer threads\n");
        return EXIT_FAILURE;
    }

    const double t_loop = stage_clock();
    double gpu_s = 0.0;
    int sent = 0;        /* captures handed to the GPU = PNG jobs started */
    int fatal = 0;
    for (int item = 0; item < argc - first_capture && !fatal; item++) {
        const double freq_mhz = atof(argv[first_capture + item]);
        printf("Frequency: %.4f MHz\n", freq_mhz);
        uint8_t *iq = NULL;
        int rows = 0;
        const int rc = capture_reader_take(&reader, item, &iq, &rows);
        if (rc > 0) {                          /* unreadable capture or short read: fatal, as in the reference -- but the */
            fatal = 1;                         /* PNGs of the captures before it are completed first (below) */
            capture_reader_release(&reader, item);
            break;
        }
        if (rc < 0) {                          /* too few transfers: reported by the reader, next capture */
            capture_reader_release(&reader, item);
            continue;
        }
        gpu_slot *g = &slot[sent & 1];
        const double t_gpu = stage_clock();
        if (g->busy) {                         /* what this slot did two captures ago: its pixels are in the writer's buffer */
            if (fsea_stream_synchronize(plan, g->stream) != 0) die("fsea_stream_synchronize");
            png_writer_submit_job(&writer, g->job, g->file_name, fft_size, g->rows);
            g->busy = 0;
        }
        if (fsea_copy_to_device_async(device, g->d_iq, iq, (size_t)rows * row_in, g->stream) != 0) die("fsea_copy_to_device_async");
        if (fsea_stream_synchronize(plan, g->stream) != 0) die("fsea_stream_synchronize");   /* (the other slot keeps running) */
        capture_reader_release(&reader, item); /* the reader may load capture item + 2 into this buffer */

        if (broad && rows >= EVALUATE_ROWS) {
            /* the first 100 rows received are the last 100 rows of the newest-first stack */
            double avg = 0.0;
            const char *oldest = (const char *)g->d_iq + (size_t)(rows - EVALUATE_ROWS) * row_in;
            if (fsea_mean_magnitude_u8_device(plan, oldest, EVALUATE_ROWS, 1, &avg, g->stream) != 0) die("gate");
            printf("\n(Average power: %.2f)\n", avg);
            if (avg < 1.1) {
                printf("Not interesting. Skipping...\n");
                gpu_s += stage_clock() - t_gpu;
                continue;
            }
        }
        uint8_t *px = png_writer_acquire_job(&writer, sent); /* waits for the PNG written from this buffer two jobs ago */
        if (fsea_exec_u8_device(plan, g->d_iq, (size_t)rows, 1, g->d_px, g->stream) != 0) die("fsea_exec_u8_device");
        if (fsea_copy_to_host_async(device, px, g->d_px, (size_t)rows * (size_t)fft_size, g->stream) != 0) die("fsea_copy_to_host_async");
        gpu_s += stage_clock() - t_gpu;
        if (broad) {
            snprintf(g->file_name, sizeof(g->file_name), "%s/broad-%.0f.png", out_dir, freq_mhz);
        } else {
            snprintf(g->file_name, sizeof(g->file_name), "%s/fft-%.4f.png", out_dir, freq_mhz);
        }
        g->rows = rows;
        g->job = sent;
        g->busy = 1;
        sent++;
    }
    /* what is still in flight, older job first */
    for (int k = 0; k < 2; k++) {
        gpu_slot *g = &slot[(sent + k) & 1];
        if (!g->busy) continue;
        const double t_gpu = stage_clock();
        if (fsea_stream_synchronize(plan, g->stream) != 0) die("fsea_stream_synchronize");
        gpu_s += stage_clock() - t_gpu;
        png_writer_submit_job(&writer, g->job, g->file_name, fft_size, g->rows);
        g->busy = 0;
    }
    if (!fatal) capture_reader_join(&reader);
    if (png_writer_finish(&writer) != 0 || fatal) return EXIT_FAILURE;
    if (timing) {
        const double wall = stage_clock() - t_loop;
        printf("Stages busy: read %.3f s, GPU (upload, gate, FFT and download queued on two streams; waits) %.3f s, PNG %.3f s; wall %.3f s for %d captures "
               "(one after the other: %.3f s)\n", reader.busy_s, gpu_s, writer.busy_s, wall, argc - first_capture,
               reader.busy_s + gpu_s + writer.busy_s);
    }
    for (int k = 0; k < 2; k++) {
        fsea_host_free(packed[k]);
        fsea_host_free(pixels[k]);
    }
    for (int k = 0; k < 2; k++) {
        fsea_device_free(device, slot[k].d_iq);
        fsea_device_free(device, slot[k].d_px);
        fsea_stream_destroy(device, slot[k].stream);
    }
    fsea_plan_destroy(plan);
    return 0;
}
