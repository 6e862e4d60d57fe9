/*
 * fsea-fft-stitch -- stitch per-frequency FFT tiles into one image (max-composite on the GPU).
 *
 * Re-statement of c/fft-stitch.c:160-189,220-224 and c/fft-stitch-broad.c:46-94:
 *   tile for frequency f is pasted at x = k * WIDTH_STEP with dst = max(dst, src),
 *   WIDTH_STEP = FFT_SIZE / (SAMPLE_RATE / FREQUENCY_STEP)  (integer division, as the reference),
 *   IMAGE_WIDTH = FFT_SIZE + FREQUENCY_RANGE * WIDTH_STEP.
 * With --footer F the image gets F extra rows carrying the frequency ruler of c/fft-stitch.c:191-217
 * (banner lines, 0.1 MHz and 1 MHz ticks, "%.2f" MHz labels; include/imgaxis.h).  --font FILE.ttf draws
 * the labels from a TrueType file as the reference does from ../fonts/RobotoCondensed-Regular.ttf
 * (c/fft-stitch.c:33); without it they are built-in dot-matrix digits.  The reference's fixed layout is
 * --rows 11211 --footer 600 --font <Roboto>.
 *
 * usage: fsea-fft-stitch [--broad] --start MHZ --end MHZ [--step MHZ] [--rows H] [--footer F] [--dir DIR]
 *                        [--device D]
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "easypng.h"
#include "fsea.h"
#include "imgaxis.h"

/* Tile PNGs are decoded ahead of the composite loop by a few threads: at the reference's own geometry (c/fft-stitch.c:16-27:
 * 300 tiles of 1024 x 16384) the inflate of one tile takes longer than its upload + max-composite on the GPU, and 300 of
 * them one after the other were two thirds of the tool's wall time (profiles/r06_reference_geometry_narrow.json).
 * Decoder t takes tiles t, t + DECODERS, ...; the main thread consumes them in tile order. */
#define DECODERS 8
typedef struct {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int full[DECODERS];
    uint8_t *tile[DECODERS];
    int w[DECODERS], h[DECODERS];
    uint32_t n_tiles;
    int broad;
    double start, step;
    const char *dir;
} tile_queue;
typedef struct {
    tile_queue *q;
    int slot;
} decoder_arg;

static void tile_name(const tile_queue *q, uint32_t k, char *file_name, size_t cap) {
    const double f = q->start + k * q->step;
    if (q->broad) snprintf(file_name, cap, "%s/broad-%.0f.png", q->dir, f);
    else snprintf(file_name, cap, "%s/fft-%.4f.png", q->dir, f);
}

static void *decoder_main(void *p) {
    const decoder_arg *a = (const decoder_arg *)p;
    tile_queue *q = a->q;
    const int s = a->slot;
    for (uint32_t k = (uint32_t)s; k < q->n_tiles; k += DECODERS) {
        char file_name[512];
        tile_name(q, k, file_name, sizeof(file_name));
        int w = 0, h = 0;
        uint8_t *tile = read_gray_png(file_name, &w, &h); /* NULL = could not load: the main thread reports it in order */
        pthread_mutex_lock(&q->mu);
        while (q->full[s]) pthread_cond_wait(&q->cv, &q->mu);
        q->tile[s] = tile;
        q->w[s] = w;
        q->h[s] = h;
        q->full[s] = 1;
        pthread_cond_broadcast(&q->cv);
        pthread_mutex_unlock(&q->mu);
    }
    return NULL;
}

static void die(const char *what) {
    fprintf(stderr, "fsea-fft-stitch: %s: %s\n", what, fsea_last_error_string());
    exit(EXIT_FAILURE);
}

int main(int argc, char **argv) {
    int broad = 0, device = 0, rows = -1, footer = 0;
    double start = -1, end = -1, step = -1;
    const char *dir = ".", *font_file = NULL;
    for (int i = 1; i < argc; i++) {
        if (strcmp(argv[i], "--broad") == 0) broad = 1;
        else if (strcmp(argv[i], "--start") == 0 && i + 1 < argc) start = atof(argv[++i]);
        else if (strcmp(argv[i], "--end") == 0 && i + 1 < argc) end = atof(argv[++i]);
        else if (strcmp(argv[i], "--step") == 0 && i + 1 < argc) step = atof(argv[++i]);
        else if (strcmp(argv[i], "--rows") == 0 && i + 1 < argc) rows = atoi(argv[++i]);
        else if (strcmp(argv[i], "--footer") == 0 && i + 1 < argc) footer = atoi(argv[++i]);
        else if (strcmp(argv[i], "--dir") == 0 && i + 1 < argc) dir = argv[++i];
        else if (strcmp(argv[i], "--device") == 0 && i + 1 < argc) device = atoi(argv[++i]);
        else if (strcmp(argv[i], "--font") == 0 && i + 1 < argc) font_file = argv[++i];
    }
    if (start < 0 || end < start) {
        fprintf(stderr, "usage: fsea-fft-stitch [--broad] --start MHZ --end MHZ [--step MHZ] [--rows H] [--dir DIR] [--footer ROWS [--font FILE.ttf]]\n");
        return EXIT_FAILURE;
    }
    const uint32_t fft_size = broad ? 256 : 1024;
    const uint32_t sample_rate = 5000000;                      /* SAMPLE_RATE */
    if (step < 0) step = broad ? 5.0 : 2.0;                     /* FREQUENCY_STEP */
    /* WIDTH_STEP = FFT_SIZE / (SAMPLE_RATE / FREQUENCY_STEP) in integers (c/fft-stitch.c:21): the step has to
     * be positive and at most the sample rate, or the divisions below are by zero */
    if (!(step > 0.0) || (uint64_t)(step * 1e6 + 0.5) > (uint64_t)sample_rate || (uint64_t)(step * 1e6 + 0.5) == 0) {
        fprintf(stderr, "ERROR: --step must be in (0, %.1f] MHz (the tiles are %.1f MHz wide)\n", sample_rate / 1e6,
                sample_rate / 1e6);
        return EXIT_FAILURE;
    }
    const uint32_t step_hz = (uint32_t)(step * 1e6 + 0.5);
    const uint32_t width_step = fft_size / (sample_rate / step_hz);
    const uint32_t n_tiles = (uint32_t)((end - start) / step + 1e-9) + 1;
    const uint32_t image_width = fft_size + (n_tiles - 1) * width_step;
    uint32_t image_height = 0;
    void *d_image = NULL, *d_tile = NULL;
    size_t tile_cap = 0;

    tile_queue queue;
    memset(&queue, 0, sizeof(queue));
    pthread_mutex_init(&queue.mu, NULL);
    pthread_cond_init(&queue.cv, NULL);
    queue.n_tiles = n_tiles;
    queue.broad = broad;
    queue.start = start;
    queue.step = step;
    queue.dir = dir;
    pthread_t decoders[DECODERS];
    decoder_arg decoder_args[DECODERS];
    for (int t = 0; t < DECODERS && (uint32_t)t < n_tiles; t++) {
        decoder_args[t].q = &queue;
        decoder_args[t].slot = t;
        if (pthread_create(&decoders[t], NULL, decoder_main, &decoder_args[t]) != 0) {
            fprintf(stderr, "fsea-fft-stitch: cannot start a decoder thread\n");
            return EXIT_FAILURE;
        }
        pthread_detach(decoders[t]); /* an error below ends the process; nothing joins them */
    }

    for (uint32_t k = 0; k < n_tiles; k++) {
        char file_name[512];
        tile_name(&queue, k, file_name, sizeof(file_name));
        printf("Composing %s...\n", file_name);
        const int slot = (int)(k % DECODERS);
        pthread_mutex_lock(&queue.mu);
        while (!queue.full[slot]) pthread_cond_wait(&queue.cv, &queue.mu);
        uint8_t *tile = queue.tile[slot];
        const int w = queue.w[slot], h = queue.h[slot];
        queue.full[slot] = 0; /* the decoder may fetch tile k + DECODERS */
        pthread_cond_broadcast(&queue.cv);
        pthread_mutex_unlock(&queue.mu);
        if (!tile) {
            fprintf(stderr, "ERROR: could not load %s\n", file_name);
            return EXIT_FAILURE;
        }
        if (image_height == 0) {
            image_height = rows > 0 ? (uint32_t)rows : (uint32_t)h;
            printf("Image size: %u x %u\n", image_width, image_height);
            if (fsea_device_alloc(device, (size_t)image_width * image_height, &d_image) != 0) die("fsea_device_alloc");
            uint8_t *zero = (uint8_t *)calloc((size_t)image_width * image_height, 1);
            if (fsea_copy_to_device(device, d_image, zero, (size_t)image_width * image_height) != 0) die("clear");
            free(zero);
        }
        if ((uint32_t)w != fft_size || (uint32_t)h < image_height) {
            fprintf(stderr, "ERROR: bad image size %s\n", file_name);
            return EXIT_FAILURE;
        }
        const size_t bytes = (size_t)fft_size * image_height;
        if (bytes > tile_cap) {
            fsea_device_free(device, d_tile);
            if (fsea_device_alloc(device, bytes, &d_tile) != 0) die("fsea_device_alloc");
            tile_cap = bytes;
        }
        if (fsea_copy_to_device(device, d_tile, tile, bytes) != 0) die("fsea_copy_to_device");
        if (fsea_composite_max_device(d_image, d_tile, k * width_step, 0, fft_size, image_height, image_width, image_height,
                                      fft_size, device, NULL) != 0) {
            die("fsea_composite_max_device");
        }
        free(tile);
    }
    const uint32_t full_height = image_height + (footer > 0 ? (uint32_t)footer : 0);
    uint8_t *image = (uint8_t *)calloc((size_t)image_width * full_height, 1);
    if (fsea_copy_to_host(device, image, d_image, (size_t)image_width * image_height) != 0) die("fsea_copy_to_host");
    if (footer > 0) {
        printf("Adding markers...\n");
        img_axis_config axis;
        memset(&axis, 0, sizeof(axis));
        axis.fft_size = fft_size;
        axis.rows = image_height;
        axis.sample_rate = sample_rate;
        axis.frequency_step = step_hz;
        axis.frequency_start = (uint64_t)(start * 1e6 + 0.5);
        axis.frequency_end = (uint64_t)(end * 1e6 + 0.5);
        axis.minor_tick_rate = 100000;
        axis.major_tick_rate = 1000000;
        axis.font_size_px = 48;
        axis.line_color = 255;
        /* FONT_FILE "../fonts/RobotoCondensed-Regular.ttf" in the reference (c/fft-stitch.c:38): the caller's file here */
        ntt_font *font = font_file ? ntt_font_load(font_file) : NULL;
        if (font_file && !font) return EXIT_FAILURE;
        axis.font = font;
        img_draw_frequency_axis(image, image_width, full_height, &axis);
        ntt_font_free(font);
    }
    char out_name[512];
    if (broad) snprintf(out_name, sizeof(out_name), "%s/broad-stitched-%.0f-%.0f.png", dir, start, end);
    else snprintf(out_name, sizeof(out_name), "%s/fft-stitched-%.4f-%.4f.png", dir, start, end);
    printf("Saving %s...\n", out_name);
    int rc = write_gray_png(out_name, (int)image_width, (int)full_height, image);
    free(image);
    fsea_device_free(device, d_tile);
    fsea_device_free(device, d_image);
    return rc == 0 ? 0 : EXIT_FAILURE;
}
