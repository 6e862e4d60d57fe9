/*
 * fsea-fft-sweep -- the reference's sweep (c/fft-batch*.c) and stitch (c/fft-stitch*.c) as ONE run over
 * the GPUs of a node.
 *
 * The reference captures one centre frequency after the other (c/fft-batch-broad.c:176-206), writes a PNG
 * tile per frequency, and a second program loads the tiles back and max-composites them side by side
 * (c/fft-stitch-broad.c:62-87, c/fft-stitch.c:160-189).  Tiles are independent, so here the list of
 * captures is cut into one contiguous range per GPU ("member"): one host thread per member with its own
 * plan and stream turns its captures into u8 dB tiles exactly as fsea-fft-batch does (same files, same
 * rows, same gate), and the finished tiles travel chunk by chunk to member 0's GPU (fsea_comm_gather:
 * RCCL grouped send/recv over xGMI, include/fsea_comm.h), where they are max-composited into the stitched
 * image while the next chunk is being computed.  Tile PNGs are still written (by the member that made
 * them) unless --no-tiles.  Every member overlaps its three host stages (pipeline.h): its next capture is being read
 * and its previous tile's PNG encoded while the current one is on the GPU.
 *
 *   default   1024-pt, 16384 rows, *10 pixels, "fft-%.4f.png",  stitched "fft-stitched-%.4f-%.4f.png"
 *   --broad    256-pt,  4096 rows, *5 pixels + DC fix + 100-row gate, "broad-%.0f.png",
 *              stitched "broad-stitched-%.0f-%.0f.png"
 * A tile the gate rejects ("Not interesting") gets no PNG and contributes nothing to the stitched image.
 *
 * usage: fsea-fft-sweep [--broad] [--devices LIST] [--rows H] [--fft N] [--skip K] [--step MHZ] [--chunk T]
 *                       [--out DIR] [--no-tiles] FREQ_MHZ=capture.raw [FREQ_MHZ=capture.raw ...]
 *   --devices 0-7 | 0,1,2 | 0,0   one member per entry (an id may repeat: members then share that GPU and
 *                                 the gather falls back to device copies)
 *   --step    frequency step between consecutive captures in MHz (default 2, --broad 5): the tile at
 *             position k is placed at x = k * FFT_SIZE / (SAMPLE_RATE / FREQUENCY_STEP)
 *   --chunk   tiles per gather (default 4)
 */
#include <fcntl.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include "easypng.h"
#include "fsea.h"
#include "fsea_comm.h"
#include "pipeline.h"

#define TRANSFER_BYTES 262144 /* one HackRF transfer: 131072 IQ samples */
#define EVALUATE_ROWS 100     /* c/fft-batch-broad.c:22 */
#define MAX_MEMBERS 64

typedef struct {
    double freq_mhz;
    const char *path;
} capture;

typedef struct {
    /* shared, read-only */
    int broad, rows_wanted, fft_size, skip, chunk, write_tiles, n_members, n_captures;
    const char *out_dir;
    const char *window; /* --window NAME: a taper on every member's plan (fsea_plan_set_window), or NULL = rectangular */
    const capture *captures;
    uint32_t width_step, image_width;
    fsea_comm *comm;
    int devices[MAX_MEMBERS];
    int lo[MAX_MEMBERS], hi[MAX_MEMBERS]; /* capture range of every member */
    void *d_image; /* the stitched image, on member 0's GPU */
} sweep_config;

typedef struct {
    const sweep_config *cfg;
    int member;
} member_arg;

/* Fatal conditions print and end the process (the reference's convention, c/fft-batch.c:41-47); a member
 * that merely returned would leave the others waiting in the next gather.  The tile PNGs that were already handed to
 * a writer thread are completed first: an error in one capture must not truncate the image of an earlier one. */
static png_writer *g_writers[MAX_MEMBERS];
static pthread_mutex_t g_fatal_mu = PTHREAD_MUTEX_INITIALIZER;
/* g_writers[] is written by its member thread and read by whichever thread fails: both under g_fatal_mu.  A
 * member that wants to retire its writer while another thread is already ending the process blocks in writer_retire --
 * it never gets to png_writer_finish, so the failing thread drains a writer whose thread is still alive. */
static void writer_register(int member, png_writer *w) {
    pthread_mutex_lock(&g_fatal_mu);
    g_writers[member] = w;
    pthread_mutex_unlock(&g_fatal_mu);
}
static void writer_retire(int member) { writer_register(member, NULL); }
static void fatal_exit(void) {
    pthread_mutex_lock(&g_fatal_mu); /* one thread ends the process; a second fatal error waits here until it has */
    /* ONE absolute deadline for all members' writers and both of their slots: the fatal path ends within about 30 s in
     * all, not 60 s per member one after the other */
    const struct timespec until = ring_deadline(30.0);
    for (int m = 0; m < MAX_MEMBERS; m++) {
        if (g_writers[m] && png_writer_drain_until(g_writers[m], &until) != 0) {
            fprintf(stderr, "fsea-fft-sweep: member %d's tile PNG was still being written 30 s after the fatal error; leaving it\n", m);
        }
    }
    exit(EXIT_FAILURE);
}
#define CHECK(ok, ...)                                                                            \
    do {                                                                                          \
        if (!(ok)) {                                                                              \
            fprintf(stderr, "fsea-fft-sweep: " __VA_ARGS__);                                      \
            fprintf(stderr, " (%s | %s)\n", fsea_last_error_string(), fsea_comm_last_error());    \
            fatal_exit();                                                                         \
        }                                                                                         \
    } while (0)

static void partition(int n_items, int world, int rank, int *lo, int *hi) {
    const int base = n_items / world, extra = n_items % world;
    *lo = rank * base + (rank < extra ? rank : extra);
    *hi = *lo + base + (rank < extra ? 1 : 0);
}

static int parse_devices(const char *text, int *devices) {
    int n = 0;
    const char *p = text;
    while (*p && n < MAX_MEMBERS) {
        char *end = NULL;
        const long a = strtol(p, &end, 10);
        if (end == p) return -1;
        long b = a;
        if (*end == '-') {
            p = end + 1;
            b = strtol(p, &end, 10);
            if (end == p || b < a) return -1;
        }
        for (long d = a; d <= b && n < MAX_MEMBERS; d++) devices[n++] = (int)d;
        p = end;
        if (*p == ',') p++;
        else if (*p) return -1;
    }
    return n;
}

/* rows of one capture, newest first: row y <- first 2N bytes of transfer skip + rows - 1 - y (c/fft-batch.c:62-74) */
static int load_capture(const sweep_config *cfg, const capture *cap, uint8_t *packed, int *rows_out) {
    const size_t row_in = (size_t)2 * (size_t)cfg->fft_size;
    /* one pread per row, as fsea-fft-batch reads (a row is the first 2N bytes of a 262144-byte transfer) */
    const int fd = open(cap->path, O_RDONLY);
    if (fd < 0) {
        fprintf(stderr, "fsea-fft-sweep: cannot open %s\n", cap->path);
        return -1;
    }
    struct stat st;
    if (fstat(fd, &st) != 0) {
        fprintf(stderr, "fsea-fft-sweep: cannot stat %s\n", cap->path);
        close(fd);
        return -1;
    }
    const long transfers = (long)(st.st_size / TRANSFER_BYTES);
    int rows = (int)(transfers - cfg->skip);
    if (rows > cfg->rows_wanted) rows = cfg->rows_wanted;
    if (rows < cfg->rows_wanted) {
        fprintf(stderr, "fsea-fft-sweep: %s holds %ld transfers, need %d after skipping %d\n", cap->path, transfers,
                cfg->rows_wanted, cfg->skip);
        close(fd);
        return -1;
    }
    for (int y = 0; y < rows; y++) {
        const off_t tr = (off_t)cfg->skip + rows - 1 - y;
        uint8_t *dst = packed + (size_t)y * row_in;
        size_t got = 0;
        while (got < row_in) {
            const ssize_t r = pread(fd, dst + got, row_in - got, tr * (off_t)TRANSFER_BYTES + (off_t)got);
            if (r <= 0) break;
            got += (size_t)r;
        }
        if (got != row_in) {
            fprintf(stderr, "Short read, samples lost, exiting!\n");
            close(fd);
            return -1;
        }
    }
    close(fd);
    *rows_out = rows;
    return 0;
}

typedef struct {
    const sweep_config *cfg;
    int first; /* index of the member's first capture */
} member_reader_ctx;

static int member_loader(void *vctx, int item, uint8_t *packed, int *rows_out) {
    const member_reader_ctx *c = (const member_reader_ctx *)vctx;
    return load_capture(c->cfg, &c->cfg->captures[c->first + item], packed, rows_out);
}

static void *member_main(void *argp) {
    member_arg *arg = (member_arg *)argp;
    const sweep_config *cfg = arg->cfg;
    const int me = arg->member, device = cfg->devices[me];
    const int n = cfg->fft_size, rows = cfg->rows_wanted;
    const size_t row_in = (size_t)2 * (size_t)n, tile_bytes = (size_t)rows * (size_t)n;
    const int mine = cfg->hi[me] - cfg->lo[me];

    fsea_plan *plan = NULL;
    /* stream: tiles with an even index, the gathers and the composites; stream2: tiles with an odd index -- consecutive
     * captures of a member go to its GPU alternately on two streams with double-buffered input (include/fsea.h) */
    void *stream = NULL, *stream2 = NULL, *d_iq2[2] = {NULL, NULL}, *d_tiles = NULL, *d_inbox = NULL;
    void *packed[2] = {NULL, NULL}, *pixels[2] = {NULL, NULL};
    for (int k = 0; k < 2; k++) {
        CHECK(fsea_host_alloc((size_t)rows * row_in, &packed[k]) == 0 && fsea_host_alloc(tile_bytes, &pixels[k]) == 0,
              "member %d: out of pinned host memory", me);
    }
    int max_tiles = 0; /* largest per-member tile count: every member takes part in that many chunk gathers */
    for (int m = 0; m < cfg->n_members; m++) {
        const int cnt = cfg->hi[m] - cfg->lo[m];
        if (cnt > max_tiles) max_tiles = cnt;
    }
    const int depth = (max_tiles + cfg->chunk - 1) / cfg->chunk;
    member_reader_ctx rctx = {cfg, cfg->lo[me]};
    capture_reader reader;
    png_writer writer;
    CHECK(capture_reader_start(&reader, mine, member_loader, &rctx, (uint8_t *)packed[0], (uint8_t *)packed[1]) == 0 &&
          png_writer_start(&writer, (uint8_t *)pixels[0], (uint8_t *)pixels[1]) == 0, "member %d: reader / writer threads", me);
    writer_register(me, &writer);
    CHECK(fsea_plan_create(&plan, n, n, cfg->broad ? FSEA_MODE_DB5_U8_DCFIX : FSEA_MODE_DB10_U8, device) == 0,
          "member %d: fsea_plan_create", me);
    if (cfg->window) {
        static const char *names[] = {"rect", "hann", "hamming", "blackman", "blackmanharris", "flattop"};
        int kind = -1;
        for (int k = 0; k < 6; k++) {
            if (strcmp(cfg->window, names[k]) == 0) kind = k;
        }
        CHECK(kind >= 0, "unknown window '%s' (hann, hamming, blackman, blackmanharris, flattop)", cfg->window);
        float *w = (float *)malloc(sizeof(float) * (size_t)n);
        CHECK(w && fsea_window_fill(kind, n, w) == 0 && (kind == 0 || fsea_plan_set_window(plan, w) == 0), "member %d: --window", me);
        free(w);
    }
    CHECK(fsea_comm_stream_create(device, &stream) == 0 && fsea_comm_stream_create(device, &stream2) == 0, "member %d: stream", me);
    CHECK(fsea_device_alloc(device, (size_t)rows * row_in, &d_iq2[0]) == 0 &&
          fsea_device_alloc(device, (size_t)rows * row_in, &d_iq2[1]) == 0, "member %d: alloc", me);
    CHECK(fsea_device_alloc(device, tile_bytes * (size_t)(mine > 0 ? mine : 1), &d_tiles) == 0, "member %d: alloc", me);
    if (me == 0) {
        CHECK(fsea_device_alloc(device, tile_bytes * (size_t)cfg->chunk * (size_t)cfg->n_members, &d_inbox) == 0,
              "root: inbox alloc");
    }

    for (int j = 0; j < depth; j++) {
        /* this member's tiles of chunk j */
        const int c_lo = j * cfg->chunk < mine ? j * cfg->chunk : mine;
        const int c_hi = (j + 1) * cfg->chunk < mine ? (j + 1) * cfg->chunk : mine;
        for (int k = c_lo; k < c_hi; k++) {
            const capture *cap = &cfg->captures[cfg->lo[me] + k];
            char *tile = (char *)d_tiles + (size_t)k * tile_bytes;
            void *const st = (k & 1) ? stream2 : stream;
            void *const d_iq = d_iq2[k & 1];
            int got_rows = 0, keep = 1;
            uint8_t *iq = NULL;
            printf("Frequency: %.4f MHz\n", cap->freq_mhz);
            CHECK(capture_reader_take(&reader, k, &iq, &got_rows) == 0, "member %d: %s", me, cap->path);
            /* this slot's stream may still read its d_iq for the tile two captures ago (the other slot keeps running) */
            CHECK(fsea_stream_synchronize(plan, st) == 0, "member %d: sync", me);
            CHECK(fsea_copy_to_device(device, d_iq, iq, (size_t)rows * row_in) == 0, "member %d: upload", me);
            capture_reader_release(&reader, k); /* the reader may load this member's capture k + 2 now */
            if (cfg->broad && rows >= EVALUATE_ROWS) {
                /* the first 100 rows received are the last 100 rows of the newest-first stack (c/fft-batch-broad.c:81-98) */
                double avg = 0.0;
                const char *oldest = (const char *)d_iq + (size_t)(rows - EVALUATE_ROWS) * row_in;
                CHECK(fsea_mean_magnitude_u8_device(plan, oldest, EVALUATE_ROWS, 1, &avg, st) == 0, "member %d: gate", me);
                printf("\n(Average power: %.2f)\n", avg);
                if (avg < 1.1) {
                    printf("Not interesting. Skipping...\n");
                    keep = 0;
                }
            }
            if (keep) {
                CHECK(fsea_exec_u8_device(plan, d_iq, (size_t)rows, 1, tile, st) == 0, "member %d: exec", me);
            } else {
                uint8_t *zero = png_writer_acquire(&writer); /* (a free pixel buffer; nothing is submitted) */
                memset(zero, 0, tile_bytes); /* an all-zero tile changes nothing under max */
                CHECK(fsea_copy_to_device(device, tile, zero, tile_bytes) == 0, "member %d: clear", me);
            }
            if (keep && cfg->write_tiles) {
                char file_name[600];
                CHECK(fsea_stream_synchronize(plan, st) == 0, "member %d: sync", me);
                uint8_t *px = png_writer_acquire(&writer); /* waits for the PNG encoded from this buffer two tiles ago */
                CHECK(fsea_copy_to_host(device, px, tile, tile_bytes) == 0, "member %d: download", me);
                if (cfg->broad) snprintf(file_name, sizeof(file_name), "%s/broad-%.0f.png", cfg->out_dir, cap->freq_mhz);
                else snprintf(file_name, sizeof(file_name), "%s/fft-%.4f.png", cfg->out_dir, cap->freq_mhz);
                png_writer_submit(&writer, file_name, n, rows);
            }
        }
        /* the chunk's odd tiles are complete before the gather reads them on `stream` */
        CHECK(fsea_stream_synchronize(plan, stream2) == 0, "member %d: sync", me);
        /* gather chunk j: member m's tiles land at slot m of the root's inbox.  The inbox is reused per
         * chunk: the transfers are queued on the root's stream behind the composites of chunk j - 1. */
        size_t bytes[MAX_MEMBERS], offsets[MAX_MEMBERS];
        for (int m = 0; m < cfg->n_members; m++) {
            const int cnt = cfg->hi[m] - cfg->lo[m];
            const int a = j * cfg->chunk < cnt ? j * cfg->chunk : cnt;
            const int b = (j + 1) * cfg->chunk < cnt ? (j + 1) * cfg->chunk : cnt;
            bytes[m] = (size_t)(b - a) * tile_bytes;
            offsets[m] = (size_t)m * (size_t)cfg->chunk * tile_bytes;
        }
        CHECK(fsea_comm_gather(cfg->comm, me, (const char *)d_tiles + (size_t)c_lo * tile_bytes, bytes, offsets, d_inbox,
                               stream) == 0,
              "member %d: gather", me);
        if (me == 0) {
            /* composite what arrived: member m's tiles of this chunk sit side by side from its first tile's x */
            for (int m = 0; m < cfg->n_members; m++) {
                const uint32_t count = (uint32_t)(bytes[m] / tile_bytes);
                if (count == 0) continue;
                const uint32_t first_tile = (uint32_t)(cfg->lo[m] + j * cfg->chunk);
                CHECK(fsea_stitch_tiles_device(cfg->d_image, (const char *)d_inbox + offsets[m], count,
                                               first_tile * cfg->width_step, cfg->width_step, (uint32_t)n, (uint32_t)rows,
                                               cfg->image_width, device, stream) == 0,
                      "root: stitch");
            }
        }
    }
    /* all transfers complete, all source tiles free again */
    CHECK(fsea_comm_barrier(cfg->comm, me, stream) == 0, "member %d: barrier", me);
    capture_reader_join(&reader);
    writer_retire(me);
    CHECK(png_writer_finish(&writer) == 0, "member %d: a tile PNG could not be written", me);
    for (int k = 0; k < 2; k++) {
        fsea_host_free(packed[k]);
        fsea_host_free(pixels[k]);
    }
    fsea_device_free(device, d_iq2[0]);
    fsea_device_free(device, d_iq2[1]);
    fsea_device_free(device, d_tiles);
    fsea_device_free(device, d_inbox);
    fsea_comm_stream_destroy(device, stream);
    fsea_comm_stream_destroy(device, stream2);
    fsea_plan_destroy(plan);
    return NULL;
}

int main(int argc, char **argv) {
    sweep_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.rows_wanted = -1;
    cfg.fft_size = -1;
    cfg.skip = 10;
    cfg.chunk = 4;
    cfg.write_tiles = 1;
    cfg.out_dir = ".";
    cfg.n_members = 1;
    double step = -1.0;
    int first_capture = argc;
    for (int i = 1; i < argc; i++) {
        if (strcmp(argv[i], "--broad") == 0) cfg.broad = 1;
        else if (strcmp(argv[i], "--no-tiles") == 0) cfg.write_tiles = 0;
        else if (strcmp(argv[i], "--window") == 0 && i + 1 < argc) cfg.window = argv[++i];
        else if (strcmp(argv[i], "--rows") == 0 && i + 1 < argc) cfg.rows_wanted = atoi(argv[++i]);
        else if (strcmp(argv[i], "--fft") == 0 && i + 1 < argc) cfg.fft_size = atoi(argv[++i]);
        else if (strcmp(argv[i], "--skip") == 0 && i + 1 < argc) cfg.skip = atoi(argv[++i]);
        else if (strcmp(argv[i], "--chunk") == 0 && i + 1 < argc) cfg.chunk = atoi(argv[++i]);
        else if (strcmp(argv[i], "--step") == 0 && i + 1 < argc) step = atof(argv[++i]);
        else if (strcmp(argv[i], "--out") == 0 && i + 1 < argc) cfg.out_dir = argv[++i];
        else if (strcmp(argv[i], "--devices") == 0 && i + 1 < argc) {
            cfg.n_members = parse_devices(argv[++i], cfg.devices);
            if (cfg.n_members <= 0) {
                fprintf(stderr, "fsea-fft-sweep: cannot parse --devices %s\n", argv[i]);
                return EXIT_FAILURE;
            }
        } else { first_capture = i; break; }
    }
    if (first_capture >= argc || cfg.chunk <= 0) {
        fprintf(stderr, "usage: fsea-fft-sweep [--broad] [--devices LIST] [--rows H] [--fft N] [--skip K] [--step MHZ] "
                        "[--chunk T] [--out DIR] [--no-tiles] [--window NAME] FREQ_MHZ=capture.raw ...\n");
        return EXIT_FAILURE;
    }
    if (cfg.fft_size < 0) cfg.fft_size = cfg.broad ? 256 : 1024;        /* FFT_SIZE */
    if (cfg.rows_wanted < 0) cfg.rows_wanted = cfg.broad ? 4096 : 16384; /* FFT_HISTORY_SIZE */
    if (step < 0) step = cfg.broad ? 5.0 : 2.0;                          /* FREQUENCY_STEP */
    if (cfg.rows_wanted < 1 || cfg.skip < 0 || cfg.fft_size < 1) {
        fprintf(stderr, "ERROR: --rows and --fft must be positive, --skip not negative\n");
        return EXIT_FAILURE;
    }
    const uint32_t sample_rate = 5000000;                                /* SAMPLE_RATE */
    if (!(step > 0.0) || (uint64_t)(step * 1e6 + 0.5) > (uint64_t)sample_rate || (uint64_t)(step * 1e6 + 0.5) == 0) {
        fprintf(stderr, "ERROR: --step must be in (0, %.1f] MHz\n", sample_rate / 1e6);
        return EXIT_FAILURE;
    }
    cfg.width_step = (uint32_t)cfg.fft_size / (sample_rate / (uint32_t)(step * 1e6 + 0.5)); /* c/fft-stitch.c:21 */

    cfg.n_captures = argc - first_capture;
    capture *captures = (capture *)calloc((size_t)cfg.n_captures, sizeof(capture));
    if (!captures) return EXIT_FAILURE;
    for (int i = 0; i < cfg.n_captures; i++) {
        char *eq = strchr(argv[first_capture + i], '=');
        if (!eq) {
            fprintf(stderr, "fsea-fft-sweep: expected FREQ_MHZ=capture.raw, got %s\n", argv[first_capture + i]);
            return EXIT_FAILURE;
        }
        captures[i].freq_mhz = atof(argv[first_capture + i]);
        captures[i].path = eq + 1;
    }
    cfg.captures = captures;
    if (cfg.n_members > cfg.n_captures) cfg.n_members = cfg.n_captures; /* no member without a capture */
    for (int m = 0; m < cfg.n_members; m++) partition(cfg.n_captures, cfg.n_members, m, &cfg.lo[m], &cfg.hi[m]);
    cfg.image_width = (uint32_t)cfg.fft_size + (uint32_t)(cfg.n_captures - 1) * cfg.width_step;
    printf("Frequency range: %.4f MHz - %.4f MHz, %d captures on %d GPU member(s)\n", captures[0].freq_mhz,
           captures[cfg.n_captures - 1].freq_mhz, cfg.n_captures, cfg.n_members);
    printf("Image size: %u x %d\n", cfg.image_width, cfg.rows_wanted);

    if (fsea_comm_create(&cfg.comm, cfg.n_members, cfg.devices) != 0) {
        fprintf(stderr, "fsea-fft-sweep: %s\n", fsea_comm_last_error());
        return EXIT_FAILURE;
    }
    printf("Gather backend: %s\n", fsea_comm_backend(cfg.comm));
    const size_t image_bytes = (size_t)cfg.image_width * (size_t)cfg.rows_wanted;
    uint8_t *image = (uint8_t *)calloc(image_bytes, 1);
    if (!image || fsea_device_alloc(cfg.devices[0], image_bytes, &cfg.d_image) != 0 ||
        fsea_copy_to_device(cfg.devices[0], cfg.d_image, image, image_bytes) != 0) {
        fprintf(stderr, "fsea-fft-sweep: image setup: %s\n", fsea_last_error_string());
        return EXIT_FAILURE;
    }

    pthread_t threads[MAX_MEMBERS];
    member_arg args[MAX_MEMBERS];
    for (int m = 0; m < cfg.n_members; m++) {
        args[m].cfg = &cfg;
        args[m].member = m;
        if (pthread_create(&threads[m], NULL, member_main, &args[m]) != 0) {
            fprintf(stderr, "fsea-fft-sweep: cannot start member thread %d\n", m);
            return EXIT_FAILURE;
        }
    }
    for (int m = 0; m < cfg.n_members; m++) pthread_join(threads[m], NULL);

    if (fsea_copy_to_host(cfg.devices[0], image, cfg.d_image, image_bytes) != 0) {
        fprintf(stderr, "fsea-fft-sweep: %s\n", fsea_last_error_string());
        return EXIT_FAILURE;
    }
    char out_name[600];
    if (cfg.broad) {
        snprintf(out_name, sizeof(out_name), "%s/broad-stitched-%.0f-%.0f.png", cfg.out_dir, captures[0].freq_mhz,
                 captures[cfg.n_captures - 1].freq_mhz);
    } else {
        snprintf(out_name, sizeof(out_name), "%s/fft-stitched-%.4f-%.4f.png", cfg.out_dir, captures[0].freq_mhz,
                 captures[cfg.n_captures - 1].freq_mhz);
    }
    printf("Saving %s...\n", out_name);
    if (write_gray_png(out_name, (int)cfg.image_width, cfg.rows_wanted, image) != 0) return EXIT_FAILURE;
    free(image);
    free(captures);
    fsea_device_free(cfg.devices[0], cfg.d_image);
    fsea_comm_destroy(cfg.comm);
    return 0;
}
