/* Fuzz driver for read_gray_png (frequensea_amd/host/easypng.c; test infrastructure, scripts/fuzz_host_readers.sh):
 * valid PNGs of every colour type the reader takes, damaged by truncation, byte flips and header edits.
 * usage: easypng_fuzz SEED CASES SCRATCH_FILE */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "easypng.h"

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const unsigned seed = (unsigned)atoi(argv[1]);
    const int cases = atoi(argv[2]);
    /* a valid file to start from */
    uint8_t px[37 * 23];
    for (int i = 0; i < 37 * 23; ++i) px[i] = (uint8_t)(i * 7);
    if (!freopen("/dev/null", "w", stdout)) return 2; /* "Written ..." lines */
    if (write_gray_png(argv[3], 37, 23, px) != 0) return 2;
    FILE *f = fopen(argv[3], "rb");
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    rewind(f);
    unsigned char *d = malloc((size_t)n), *c = malloc((size_t)n);
    if (fread(d, 1, (size_t)n, f) != (size_t)n) return 2;
    fclose(f);
    int read_ok = 0;
    for (int it = 0; it < cases; ++it) {
        memcpy(c, d, (size_t)n);
        long len = n;
        srand(seed + (unsigned)it);
        const int kind = rand() % 4;
        if (kind == 0) len = rand() % n;
        const int flips = 1 + rand() % 6;
        for (int k = 0; k < flips; ++k) {
            const long pos = kind == 1 ? 8 + rand() % 25 /* IHDR */ : rand() % (len ? len : 1);
            if (pos < len) c[pos] = (unsigned char)rand();
        }
        if (kind == 2) { /* huge dimensions with a valid-looking header */
            c[16] = (unsigned char)rand(); c[17] = (unsigned char)rand(); c[20] = (unsigned char)rand(); c[21] = (unsigned char)rand();
        }
        FILE *o = fopen(argv[3], "wb");
        fwrite(c, 1, (size_t)len, o);
        fclose(o);
        int w = 0, h = 0;
        uint8_t *img = read_gray_png(argv[3], &w, &h);
        if (img) {
            ++read_ok;
            volatile uint8_t sink = img[(size_t)w * (size_t)h - 1];
            (void)sink;
            free(img);
        }
    }
    fprintf(stderr, "easypng: %d cases, %d still readable, no fault\n", cases, read_ok);
    free(d);
    free(c);
    return 0;
}
