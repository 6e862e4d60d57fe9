/*
 * fsea-add-markers -- header, footer, frequency ticks and labels around a stitched broad sweep image.
 *
 * Re-statement of /root/reference/c/add-markers.c:143-243: reads broad-stitched-START-END.png (what
 * fsea-fft-stitch --broad / fsea-fft-sweep --broad write), places it between a 300-row header and a
 * 300-row footer, draws the border lines, 1 MHz minor and 50 MHz major ticks at absolute frequencies and
 * a "%.2f" label per major tick (include/imgaxis.h), writes broad-stitched-START-END-markers.png.
 * The reference asserts one poster size (23693 x 7157); here the image size is whatever the stitched
 * file has.  Host-only: no GPU work in this step.
 *
 * usage: fsea-add-markers --start MHZ --end MHZ [--dir DIR] [--header H] [--footer F] [--major MHZ] [--minor MHZ]
 *                         [--font FILE.ttf]   (TrueType labels; default: built-in dot-matrix digits)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "easypng.h"
#include "imgaxis.h"

int main(int argc, char **argv) {
    double start = -1, end = -1, major = 50.0, minor = 1.0;
    int header = 300, footer = 300;
    const char *dir = ".", *font_file = NULL;
    for (int i = 1; i < argc; i++) {
        if (strcmp(argv[i], "--start") == 0 && i + 1 < argc) start = atof(argv[++i]);
        else if (strcmp(argv[i], "--end") == 0 && i + 1 < argc) end = atof(argv[++i]);
        else if (strcmp(argv[i], "--dir") == 0 && i + 1 < argc) dir = argv[++i];
        else if (strcmp(argv[i], "--header") == 0 && i + 1 < argc) header = atoi(argv[++i]);
        else if (strcmp(argv[i], "--footer") == 0 && i + 1 < argc) footer = atoi(argv[++i]);
        else if (strcmp(argv[i], "--major") == 0 && i + 1 < argc) major = atof(argv[++i]);
        else if (strcmp(argv[i], "--minor") == 0 && i + 1 < argc) minor = atof(argv[++i]);
        else if (strcmp(argv[i], "--font") == 0 && i + 1 < argc) font_file = argv[++i];
    }
    if (start < 0 || end < start || header < 10 || footer < 120 || !(major > 0) || !(minor > 0)) {
        fprintf(stderr, "usage: fsea-add-markers --start MHZ --end MHZ [--dir DIR] [--header H>=10] [--footer F>=120] "
                        "[--major MHZ] [--minor MHZ] [--font FILE.ttf]\n");
        return EXIT_FAILURE;
    }
    printf("Frequency range: %.0f MHz - %.0f MHz\n", start, end);
    char in_name[600], out_name[600];
    snprintf(in_name, sizeof(in_name), "%s/broad-stitched-%.0f-%.0f.png", dir, start, end);
    snprintf(out_name, sizeof(out_name), "%s/broad-stitched-%.0f-%.0f-markers.png", dir, start, end);
    printf("Reading %s...\n", in_name);
    int width = 0, height = 0;
    uint8_t *in = read_gray_png(in_name, &width, &height);
    if (!in) {
        fprintf(stderr, "ERROR: could not load %s\n", in_name);
        return EXIT_FAILURE;
    }
    const int out_height = height + header + footer;
    uint8_t *out = (uint8_t *)calloc((size_t)width * (size_t)out_height, 1);
    if (!out) return EXIT_FAILURE;
    printf("Composing...\n");
    for (int y = 0; y < height; y++) {
        memcpy(out + (size_t)(header + y) * (size_t)width, in + (size_t)y * (size_t)width, (size_t)width);
    }
    printf("Adding markers...\n");
    img_markers_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.source_height = (uint32_t)height;
    cfg.header_height = (uint32_t)header;
    cfg.footer_height = (uint32_t)footer;
    cfg.footer_bleed = 35;
    cfg.sample_rate = 5000000;
    cfg.frequency_start = (uint64_t)(start * 1e6 + 0.5);
    cfg.frequency_end = (uint64_t)(end * 1e6 + 0.5);
    cfg.minor_tick_rate = (uint32_t)(minor * 1e6 + 0.5);
    cfg.minor_tick_height = 30;
    cfg.major_tick_rate = (uint32_t)(major * 1e6 + 0.5);
    cfg.major_tick_height = 60;
    cfg.font_size_px = 64;
    cfg.line_color = 255;
    /* FONT_FILE "../fonts/RobotoCondensed-Bold.ttf" in the reference (c/add-markers.c:168): the caller's file here */
    ntt_font *font = font_file ? ntt_font_load(font_file) : NULL;
    if (font_file && !font) return EXIT_FAILURE;
    cfg.font = font;
    img_draw_broad_markers(out, (uint32_t)width, &cfg);
    ntt_font_free(font);
    printf("Writing %s...\n", out_name);
    if (write_gray_png(out_name, width, out_height, out) != 0) return EXIT_FAILURE;
    free(in);
    free(out);
    return 0;
}
