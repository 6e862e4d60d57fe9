/*
 * imgaxis.c -- frequency ruler of the stitched sweep image (include/imgaxis.h).
 *
 * Reference behaviour restated (paths under /root/reference): c/fft-stitch.c:56-72 (pixel, vline,
 * hline) and :191-217 (banner, ticks, labels).  The tick positions are doubles stepped by a
 * fractional pixel count and truncated when passed as uint32_t, exactly as the reference does.
 */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "imgaxis.h"

void img_pixel_put(uint8_t *buffer, uint32_t stride, uint32_t height, uint32_t x, uint32_t y, uint8_t v) {
    if (x > 0 && y > 0 && x < stride && y < height) buffer[(size_t)y * stride + x] = v;
}

void img_vline(uint8_t *buffer, uint32_t stride, uint32_t height, uint32_t x1, uint32_t y1, uint32_t y2, uint8_t v) {
    for (uint32_t y = y1; y < y2; y++) img_pixel_put(buffer, stride, height, x1, y, v);
}

void img_hline(uint8_t *buffer, uint32_t stride, uint32_t height, uint32_t x1, uint32_t y1, uint32_t x2, uint8_t v) {
    for (uint32_t x = x1; x < x2; x++) img_pixel_put(buffer, stride, height, x, y1, v);
}

/* 5 x 7 dot matrix, one byte per row, bit 4 = leftmost column */
static const uint8_t GLYPHS[12][7] = {
    {0x0e, 0x11, 0x13, 0x15, 0x19, 0x11, 0x0e}, /* 0 */
    {0x04, 0x0c, 0x04, 0x04, 0x04, 0x04, 0x0e}, /* 1 */
    {0x0e, 0x11, 0x01, 0x02, 0x04, 0x08, 0x1f}, /* 2 */
    {0x1f, 0x02, 0x04, 0x02, 0x01, 0x11, 0x0e}, /* 3 */
    {0x02, 0x06, 0x0a, 0x12, 0x1f, 0x02, 0x02}, /* 4 */
    {0x1f, 0x10, 0x1e, 0x01, 0x01, 0x11, 0x0e}, /* 5 */
    {0x06, 0x08, 0x10, 0x1e, 0x11, 0x11, 0x0e}, /* 6 */
    {0x1f, 0x01, 0x02, 0x04, 0x08, 0x08, 0x08}, /* 7 */
    {0x0e, 0x11, 0x11, 0x0e, 0x11, 0x11, 0x0e}, /* 8 */
    {0x0e, 0x11, 0x11, 0x0f, 0x01, 0x02, 0x0c}, /* 9 */
    {0x00, 0x00, 0x00, 0x00, 0x00, 0x0c, 0x0c}, /* . */
    {0x00, 0x00, 0x00, 0x1f, 0x00, 0x00, 0x00}, /* - */
};

static int glyph_index(char c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c == '.') return 10;
    if (c == '-') return 11;
    return -1;
}

int img_draw_text(uint8_t *buffer, uint32_t image_width, uint32_t image_height, const char *text, int x, int y,
                  int height_px, uint8_t v) {
    if (x < 0) return 0; /* c/fft-stitch.c:130 */
    const int cell = height_px / 7 > 0 ? height_px / 7 : 1;
    int pen = 0;
    for (const char *p = text; *p; p++) {
        const int g = glyph_index(*p);
        if (g >= 0) {
            for (int gy = 0; gy < 7; gy++) {
                for (int gx = 0; gx < 5; gx++) {
                    if (!(GLYPHS[g][gy] & (0x10 >> gx))) continue;
                    for (int sy = 0; sy < cell; sy++) {
                        for (int sx = 0; sx < cell; sx++) {
                            const int px = x + pen + gx * cell + sx, py = y + gy * cell + sy;
                            if (px < 0 || py < 0 || (uint32_t)px >= image_width || (uint32_t)py >= image_height) continue;
                            uint8_t *d = buffer + (size_t)py * image_width + (size_t)px;
                            if (*d < v) *d = v;
                        }
                    }
                }
            }
        }
        pen += 6 * cell;
    }
    return pen;
}

int img_draw_frequency_axis(uint8_t *buffer, uint32_t image_width, uint32_t image_height,
                            const img_axis_config *cfg) {
    if (cfg->rows >= image_height) return 0;
    const uint32_t footer_height = image_height - cfg->rows;
    /* c/fft-stitch.c:27-31: pixels per tick */
    const double px_per_hz = cfg->fft_size / (double)cfg->frequency_step / 2;
    const double minor_tick_size = px_per_hz * cfg->minor_tick_rate;
    const double major_tick_size = px_per_hz * cfg->major_tick_rate;
    const uint32_t markers_y = cfg->rows + (footer_height / 2 - cfg->font_size_px / 2);

    uint32_t banner_y = cfg->rows;
    uint32_t banner_bottom = image_height;
    for (int i = 0; i < 10; i++) {
        img_hline(buffer, image_width, image_height, 0, banner_y++, image_width, cfg->line_color);
        img_hline(buffer, image_width, image_height, 0, banner_bottom--, image_width, cfg->line_color);
    }
    banner_bottom++;

    if (minor_tick_size > 0) {
        for (double x = 0; x < image_width; x += minor_tick_size) {
            img_vline(buffer, image_width, image_height, (uint32_t)x, banner_y, banner_y + 50, cfg->line_color);
            img_vline(buffer, image_width, image_height, (uint32_t)x, banner_bottom - 50, banner_bottom, cfg->line_color);
        }
    }

    int labelled = 0;
    if (major_tick_size > 0) {
        long long freq = (long long)cfg->frequency_start - (cfg->sample_rate / 2) + (cfg->major_tick_rate / 2);
        const double start_x = cfg->fft_size / (double)cfg->sample_rate * (cfg->major_tick_rate / 2);
        for (double x = start_x; x < image_width; x += major_tick_size) {
            img_vline(buffer, image_width, image_height, (uint32_t)x, banner_y, banner_y + 100, cfg->line_color);
            img_vline(buffer, image_width, image_height, (uint32_t)x, banner_bottom - 100, banner_bottom, cfg->line_color);
            if (freq >= 0 && freq < (long long)cfg->frequency_end + (cfg->sample_rate / 2)) {
                if (cfg->font_size_px > 0) {
                    char text[64];
                    snprintf(text, sizeof(text), "%.2f", (double)freq / 1e6);
                    if (cfg->font != NULL) {
                        ntt_font_draw(cfg->font, buffer, image_width, image_height, text, (int)x, (int)markers_y,
                                      (int)cfg->font_size_px);
                    } else {
                        img_draw_text(buffer, image_width, image_height, text, (int)x, (int)markers_y,
                                      (int)cfg->font_size_px, cfg->line_color);
                    }
                }
                labelled++;
            }
            freq += cfg->major_tick_rate;
        }
    }
    return labelled;
}

/* c/add-markers.c:136-141: position of an absolute frequency on the poster; frequencies left of the
 * image are skipped (the reference's unsigned subtraction wraps there and fails its `x > 0 && x < width`
 * test by way of an out-of-range double -> int conversion) */
static int markers_frequency_to_x(uint32_t image_width, uint64_t real_start, uint64_t real_range, uint64_t freq) {
    if (freq < real_start) return -1;
    return (int)round((double)(freq - real_start) / (double)real_range * image_width);
}

int img_draw_broad_markers(uint8_t *buffer, uint32_t image_width, const img_markers_config *cfg) {
    const uint32_t out_height = cfg->header_height + cfg->source_height + cfg->footer_height;
    const uint64_t real_start = cfg->frequency_start - (cfg->sample_rate / 2);
    const uint64_t real_end = cfg->frequency_end + (cfg->sample_rate / 2);
    const uint64_t real_range = real_end - real_start;
    const uint32_t header_bottom = cfg->header_height;
    uint32_t footer_top = cfg->header_height + cfg->source_height;
    const uint32_t footer_bottom = out_height - 1;
    /* c/add-markers.c:196-201: white lines at header bottom + footer top */
    for (uint32_t i = 0; i < 10; i++) {
        img_hline(buffer, image_width, out_height, 0, header_bottom - i, image_width, cfg->line_color);
        img_hline(buffer, image_width, out_height, 0, footer_top + i, image_width, cfg->line_color);
    }
    footer_top += 10;
    /* :203-212 minor ticks, three pixels wide */
    if (cfg->minor_tick_rate > 0) {
        for (uint64_t freq = 0; freq < real_end; freq += cfg->minor_tick_rate) {
            const int x = markers_frequency_to_x(image_width, real_start, real_range, freq);
            if (x > 0 && x < (int)image_width) {
                for (int dx = -1; dx <= 1; dx++) {
                    img_vline(buffer, image_width, out_height, (uint32_t)(x + dx), footer_top, footer_top + cfg->minor_tick_height,
                              cfg->line_color);
                    img_vline(buffer, image_width, out_height, (uint32_t)(x + dx),
                              footer_bottom - cfg->minor_tick_height - cfg->footer_bleed, footer_bottom + 1, cfg->line_color);
                }
            }
        }
    }
    /* :214-230 major ticks, five pixels wide, and their labels */
    int labelled = 0;
    const uint32_t labels_y = cfg->source_height + cfg->header_height +
                              (cfg->footer_height / 2 - cfg->font_size_px / 2 - cfg->footer_bleed / 2);
    if (cfg->major_tick_rate > 0) {
        for (uint64_t freq = 0; freq < real_end; freq += cfg->major_tick_rate) {
            const int x = markers_frequency_to_x(image_width, real_start, real_range, freq);
            if (x > 0 && x < (int)image_width) {
                for (int dx = -2; dx <= 2; dx++) {
                    img_vline(buffer, image_width, out_height, (uint32_t)(x + dx), footer_top, footer_top + cfg->major_tick_height,
                              cfg->line_color);
                    img_vline(buffer, image_width, out_height, (uint32_t)(x + dx),
                              footer_bottom - cfg->major_tick_height - cfg->footer_bleed, footer_bottom + 1, cfg->line_color);
                }
                if (freq > real_start && freq < real_end) {
                    if (cfg->font_size_px > 0) {
                        char text[64];
                        snprintf(text, sizeof(text), "%.2f", (double)freq / 1e6);
                        if (cfg->font != NULL) {
                            ntt_font_draw(cfg->font, buffer, image_width, out_height, text, x, (int)labels_y,
                                          (int)cfg->font_size_px);
                        } else {
                            img_draw_text(buffer, image_width, out_height, text, x, (int)labels_y, (int)cfg->font_size_px,
                                          cfg->line_color);
                        }
                    }
                    labelled++;
                }
            }
        }
    }
    return labelled;
}
