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
 * The per-sample loops and FFTW are one fused GPU launch per centre frequency (include/fsea.h).
 *
 * usage: fsea-fft-batch [--broad] [--rows H] [--fft N] [--skip K] [--out DIR] [--device D]
 *                       FREQ_MHZ=capture.raw [FREQ_MHZ=capture.raw ...]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "easypng.h"
#include "fsea.h"

#define TRANSFER_BYTES 262144 /* one HackRF transfer: 131072 IQ samples */
#define EVALUATE_ROWS 100     /* c/fft-batch-broad.c:22 */

static void die(const char *what) {
    fprintf(stderr, "fsea-fft-batch: %s: %s\n", what, fsea_last_error_string());
    exit(EXIT_FAILURE);
}

int main(int argc, char **argv) {
    int broad = 0, rows_wanted = -1, fft_size = -1, skip = 10, device = 0;
    const char *out_dir = ".";
    int first_capture = argc;
    for (int i = 1; i < argc; i++) {
        if (strcmp(argv[i], "--broad") == 0) broad = 1;
        else if (strcmp(argv[i], "--rows") == 0 && i + 1 < argc) rows_wanted = atoi(argv[++i]);
        else if (strcmp(argv[i], "--fft") == 0 && i + 1 < argc) fft_size = atoi(argv[++i]);
        else if (strcmp(argv[i], "--skip") == 0 && i + 1 < argc) skip = atoi(argv[++i]);
        else if (strcmp(argv[i], "--out") == 0 && i + 1 < argc) out_dir = argv[++i];
        else if (strcmp(argv[i], "--device") == 0 && i + 1 < argc) device = atoi(argv[++i]);
        else { first_capture = i; break; }
    }
    if (first_capture >= argc) {
        fprintf(stderr, "usage: fsea-fft-batch [--broad] [--rows H] [--fft N] [--skip K] [--out DIR] "
                        "[--device D] FREQ_MHZ=capture.raw ...\n");
        return EXIT_FAILURE;
    }
    if (fft_size < 0) fft_size = broad ? 256 : 1024;          /* FFT_SIZE */
    if (rows_wanted < 0) rows_wanted = broad ? 4096 : 16384;   /* FFT_HISTORY_SIZE */
    const size_t row_in = (size_t)2 * (size_t)fft_size;

    fsea_plan *plan = NULL;
    if (fsea_plan_create(&plan, fft_size, fft_size, broad ? FSEA_MODE_DB5_U8_DCFIX : FSEA_MODE_DB10_U8, device) != 0) {
        die("fsea_plan_create");
    }
    void *d_iq = NULL, *d_px = NULL;
    if (fsea_device_alloc(device, (size_t)rows_wanted * row_in, &d_iq) != 0) die("fsea_device_alloc");
    if (fsea_device_alloc(device, (size_t)rows_wanted * (size_t)fft_size, &d_px) != 0) die("fsea_device_alloc");
    uint8_t *packed = (uint8_t *)malloc((size_t)rows_wanted * row_in);
    uint8_t *pixels = (uint8_t *)malloc((size_t)rows_wanted * (size_t)fft_size);

    for (int i = first_capture; i < argc; i++) {
        char *eq = strchr(argv[i], '=');
        if (!eq) {
            fprintf(stderr, "fsea-fft-batch: expected FREQ_MHZ=capture.raw, got %s\n", argv[i]);
            return EXIT_FAILURE;
        }
        const double freq_mhz = atof(argv[i]);
        const char *path = eq + 1;
        printf("Frequency: %.4f MHz\n", freq_mhz);
        FILE *fp = fopen(path, "rb");
        if (!fp) {
            fprintf(stderr, "fsea-fft-batch: cannot open %s\n", path);
            return EXIT_FAILURE;
        }
        fseek(fp, 0L, SEEK_END);
        const long transfers = ftell(fp) / TRANSFER_BYTES;
        int rows = (int)(transfers - skip);
        if (rows > rows_wanted) rows = rows_wanted;
        if (rows <= 0) {
            fprintf(stderr, "fsea-fft-batch: %s holds %ld transfers, need more than %d\n", path, transfers, skip);
            fclose(fp);
            continue;
        }
        /* row y (newest first) <- first 2N bytes of transfer skip + rows - 1 - y */
        for (int y = 0; y < rows; y++) {
            const long tr = (long)skip + rows - 1 - y;
            fseek(fp, tr * (long)TRANSFER_BYTES, SEEK_SET);
            if (fread(packed + (size_t)y * row_in, 1, row_in, fp) != row_in) {
                fprintf(stderr, "Short read, samples lost, exiting!\n");
                return EXIT_FAILURE;
            }
        }
        fclose(fp);
        if (fsea_copy_to_device(device, d_iq, packed, (size_t)rows * row_in) != 0) die("fsea_copy_to_device");

        if (broad && rows >= EVALUATE_ROWS) {
            /* the first 100 rows received are the last 100 rows of the newest-first stack */
            double avg = 0.0;
            const char *oldest = (const char *)d_iq + (size_t)(rows - EVALUATE_ROWS) * row_in;
            if (fsea_mean_magnitude_u8_device(plan, oldest, EVALUATE_ROWS, 1, &avg, NULL) != 0) die("gate");
            printf("\n(Average power: %.2f)\n", avg);
            if (avg < 1.1) {
                printf("Not interesting. Skipping...\n");
                continue;
            }
        }
        if (fsea_exec_u8_device(plan, d_iq, (size_t)rows, 1, d_px, NULL) != 0) die("fsea_exec_u8_device");
        if (fsea_copy_to_host(device, pixels, d_px, (size_t)rows * (size_t)fft_size) != 0) die("fsea_copy_to_host");
        char file_name[512];
        if (broad) {
            snprintf(file_name, sizeof(file_name), "%s/broad-%.0f.png", out_dir, freq_mhz);
        } else {
            snprintf(file_name, sizeof(file_name), "%s/fft-%.4f.png", out_dir, freq_mhz);
        }
        if (write_gray_png(file_name, fft_size, rows, pixels) != 0) return EXIT_FAILURE;
    }
    free(packed);
    free(pixels);
    fsea_device_free(device, d_iq);
    fsea_device_free(device, d_px);
    fsea_plan_destroy(plan);
    return 0;
}
