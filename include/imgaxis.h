/*
 * imgaxis.h -- the frequency ruler under a stitched FFT sweep image.
 *
 * Interface being replaced (paths under /root/reference; static functions and main() code of the
 * stitch tool, restated as a small library so that it can be tested without a GPU):
 *   c/fft-stitch.c:56-72     img_pixel_put / img_vline / img_hline
 *   c/fft-stitch.c:191-217   banner lines, minor + major ticks, one "%.2f" MHz label per major tick
 * Labels: the reference rasterises RobotoCondensed-Regular.ttf with stb_truetype.  With a font loaded
 * (ntt_font_load, include/ntt_font.h; the file is the user's, none is shipped) the same strings are
 * drawn by this repository's own TrueType rasteriser with the reference's positioning (centred on the
 * tick, baseline from the font's ascent); without one, a built-in 5x7 dot-matrix digit font scaled to
 * the requested pixel height is used.  Lines and ticks are pixel-identical to the reference's.
 */
#ifndef FSEA_IMGAXIS_H
#define FSEA_IMGAXIS_H

#include <stdint.h>

#include "ntt_font.h"

/* One pixel; like the reference, column 0 and row 0 are never written (its guard is
 * `x > 0 && y > 0`, c/fft-stitch.c:56-60).  Additionally clipped to the image. */
void img_pixel_put(uint8_t *buffer, uint32_t stride, uint32_t height, uint32_t x, uint32_t y, uint8_t v);
/* x = x1, y in [y1, y2) */
void img_vline(uint8_t *buffer, uint32_t stride, uint32_t height, uint32_t x1, uint32_t y1, uint32_t y2, uint8_t v);
/* y = y1, x in [x1, x2) */
void img_hline(uint8_t *buffer, uint32_t stride, uint32_t height, uint32_t x1, uint32_t y1, uint32_t x2, uint8_t v);

typedef struct {
    uint32_t fft_size;         /* FFT_SIZE: tile width in bins */
    uint32_t rows;             /* FFT_HISTORY_SIZE: first row of the footer */
    uint32_t sample_rate;      /* SAMPLE_RATE, Hz */
    uint32_t frequency_step;   /* FREQUENCY_STEP between tiles, Hz */
    uint64_t frequency_start;  /* centre frequency of the first tile, Hz */
    uint64_t frequency_end;    /* centre frequency of the last tile, Hz */
    uint32_t minor_tick_rate;  /* 0.1e6 in the reference */
    uint32_t major_tick_rate;  /* 1e6 */
    uint32_t font_size_px;     /* 48; 0 = no labels */
    uint8_t line_color;        /* 255 */
    const ntt_font *font;      /* labels drawn with ntt_font_draw, centred on the tick as the reference does
                                * (c/fft-stitch.c:214); NULL = the built-in dot-matrix digits, left-aligned */
} img_axis_config;

/* Draws the ruler of c/fft-stitch.c:191-217 into rows [cfg->rows, image_height) of
 * buffer[image_height][image_width]: ten banner lines at the top and bottom of the footer,
 * minor ticks 50 px and major ticks 100 px long from both banners, and the labels.
 * Returns the number of major ticks that received a label. */
int img_draw_frequency_axis(uint8_t *buffer, uint32_t image_width, uint32_t image_height,
                            const img_axis_config *cfg);

/* Header + footer of the broad sweep poster (c/add-markers.c:143-243): the stitched image sits between a
 * header and a footer; ten border lines close the header at its bottom and open the footer at its top;
 * the footer carries minor ticks (3 px wide) and major ticks (5 px wide) at absolute frequencies, from its
 * top edge downwards and from its bottom edge (minus the print bleed) upwards, and a "%.2f" MHz label per
 * major tick inside the covered range.  The reference hard-codes one poster size (23693 x 7157) and
 * asserts it; here the geometry comes from the caller. */
typedef struct {
    uint32_t source_height;    /* rows of the stitched image placed at row header_height */
    uint32_t header_height;    /* HEADER_HEIGHT 300 */
    uint32_t footer_height;    /* FOOTER_HEIGHT 300 */
    uint32_t footer_bleed;     /* FOOTER_BLEED 35 */
    uint32_t sample_rate;      /* SAMPLE_RATE 5e6: the image spans start - rate/2 ... end + rate/2 */
    uint64_t frequency_start;  /* centre frequency of the first tile, Hz */
    uint64_t frequency_end;    /* centre frequency of the last tile, Hz */
    uint32_t minor_tick_rate;  /* 1e6 */
    uint32_t minor_tick_height;/* 30 */
    uint32_t major_tick_rate;  /* 50e6 */
    uint32_t major_tick_height;/* 60 */
    uint32_t font_size_px;     /* 64; 0 = no labels */
    uint8_t line_color;        /* 255 */
    const ntt_font *font;      /* as in img_axis_config (c/add-markers.c:227) */
} img_markers_config;

/* Draws borders, ticks and labels into buffer[header + source + footer][image_width] (the stitched image
 * must already be in place; its rows are not touched except by the header's border line at row
 * header_height, as in the reference).  Returns the number of labels drawn. */
int img_draw_broad_markers(uint8_t *buffer, uint32_t image_width, const img_markers_config *cfg);

/* Dot-matrix text (digits, '.', '-'): each font cell is scaled to height_px rows; (x, y) is the
 * top-left corner.  Pixels are max-composited like the reference's glyph bitmaps
 * (c/fft-stitch.c:151).  Returns the advance in pixels. */
int img_draw_text(uint8_t *buffer, uint32_t image_width, uint32_t image_height, const char *text, int x, int y,
                  int height_px, uint8_t v);

#endif
