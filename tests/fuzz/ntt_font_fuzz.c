/* Fuzz driver for frequensea_amd/host/ntt_font.c (test infrastructure; built by scripts/fuzz_host_readers.sh under
 * AddressSanitizer + UndefinedBehaviorSanitizer): truncations and byte flips of a real font file, every surviving
 * load is measured and drawn at two sizes.
 * usage: ntt_font_fuzz FONT.ttf SEED CASES SCRATCH_FILE */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ntt_font.h"

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    rewind(f);
    unsigned char *d = malloc((size_t)n), *c = malloc((size_t)n);
    uint8_t *img = malloc(300 * 80);
    if (!d || !c || !img || fread(d, 1, (size_t)n, f) != (size_t)n) return 2;
    fclose(f);
    const unsigned seed = (unsigned)atoi(argv[2]);
    const int cases = atoi(argv[3]);
    int loaded = 0;
    for (int it = 0; it < cases; ++it) {
        memcpy(c, d, (size_t)n);
        long len = n;
        srand(seed + (unsigned)it);
        const int kind = rand() % 6;
        if (kind == 0) len = rand() % n; /* truncation */
        int flips = 1 + rand() % 8;
        /* kinds 4, 5: aimed at one table (its first bytes, or many bytes anywhere in it) found through the directory */
        long t_off = 0, t_len = 0;
        if (kind >= 4) {
            static const char *tags[] = {"head", "hhea", "maxp", "cmap", "loca", "hmtx", "kern", "glyf"};
            const char *tag = tags[rand() % 8];
            const int n_tab = (d[4] << 8) | d[5];
            for (int t = 0; t < n_tab && 12 + 16 * (long)t + 16 <= n; ++t) {
                const unsigned char *rec = d + 12 + 16 * t;
                if (memcmp(rec, tag, 4) == 0) {
                    t_off = ((long)rec[8] << 24) | (rec[9] << 16) | (rec[10] << 8) | rec[11];
                    t_len = ((long)rec[12] << 24) | (rec[13] << 16) | (rec[14] << 8) | rec[15];
                }
            }
            if (t_off + t_len > n) t_len = 0;
            if (kind == 5) flips = 50 + rand() % 200;
        }
        for (int k = 0; k < flips; ++k) {
            long pos;
            if (kind == 1) pos = rand() % 400;                                             /* the table directory */
            else if (kind == 4 && t_len) pos = t_off + rand() % (t_len < 600 ? t_len : 600);  /* a table's header */
            else if (kind == 5 && t_len) pos = t_off + (long)((double)rand() / RAND_MAX * (double)(t_len - 1));
            else pos = (long)((double)rand() / RAND_MAX * (double)(len ? len - 1 : 0));
            if (pos < len) c[pos] = (unsigned char)rand();
        }
        FILE *o = fopen(argv[4], "wb");
        if (!o) return 2;
        fwrite(c, 1, (size_t)len, o);
        fclose(o);
        ntt_font *font = ntt_font_load(argv[4]);
        if (!font) continue;
        ++loaded;
        int w, h;
        ntt_font_measure(font, "0123456789.-AgWij%", 0, 0, 48, &w, &h);
        memset(img, 0, 300 * 80);
        ntt_font_draw(font, img, 300, 80, "0123456789.-Ag", 150, 10, 48);
        ntt_font_draw(font, img, 300, 80, "88", 150, 10, 300);
        ntt_font_free(font);
    }
    printf("%s: %d cases, %d loaded, no fault\n", argv[1], cases, loaded);
    free(d);
    free(c);
    free(img);
    return 0;
}
