/*
 * ntt_font.h -- TrueType labels for the stitched-image rulers.
 *
 * Interface being replaced (static helpers of the reference's tools, paths under /root/reference):
 *   c/fft-stitch.c:76-94, c/add-markers.c:45-63     ntt_font + ntt_font_load
 *   c/fft-stitch.c:97-124, c/add-markers.c:66-93    ntt_font_measure
 *   c/fft-stitch.c:126-157, c/add-markers.c:95-133  ntt_font_draw
 * Same names, argument meaning and pixel conventions (string centred on x, baseline =
 * (int)(ascent * scale), per-glyph truncated advances, max-composited coverage); the rasteriser
 * underneath is this repository's own (frequensea_amd/host/ntt_font.c), not stb_truetype.  The image
 * height is an extra argument of ntt_font_draw so that glyphs are clipped instead of written past the
 * buffer.  ntt_font_load returns NULL (with a message) when the file cannot be used.
 */
#ifndef FSEA_NTT_FONT_H
#define FSEA_NTT_FONT_H

#include <stdint.h>

typedef struct ntt_font ntt_font;

ntt_font *ntt_font_load(const char *font_file);
void ntt_font_free(ntt_font *font);
void ntt_font_measure(const ntt_font *font, const char *text, const int x, const int y, const int font_size, int *width,
                      int *height);
void ntt_font_draw(const ntt_font *font, uint8_t *img, const uint32_t img_stride, const uint32_t img_height, const char *text,
                   const int x, const int y, const int font_size);

/* The pieces the helpers are made of (what the reference takes from stb_truetype: stbtt_FindGlyphIndex,
 * ScaleForPixelHeight, GetFontVMetrics, GetGlyphHMetrics, GetGlyphKernAdvance, GetGlyphBox,
 * GetGlyphBitmapBox, GetGlyphBitmap); font units unless a scale is passed. */
int ntt_font_glyph_index(const ntt_font *font, int codepoint);
float ntt_font_scale_for_pixel_height(const ntt_font *font, float pixels);
void ntt_font_vmetrics(const ntt_font *font, int *ascent, int *descent, int *line_gap);
void ntt_font_hmetrics(const ntt_font *font, int glyph, int *advance, int *lsb);
int ntt_font_kern_advance(const ntt_font *font, int glyph1, int glyph2);
int ntt_font_glyph_box(const ntt_font *font, int glyph, int *x0, int *y0, int *x1, int *y1);
void ntt_font_bitmap_box(const ntt_font *font, int glyph, float scale, int *ix0, int *iy0, int *ix1, int *iy1);
/* malloc'd width x height coverage (0..255), top-left at (xoff, yoff) relative to the pen on the baseline;
 * NULL for empty or composite glyphs */
uint8_t *ntt_font_glyph_bitmap(const ntt_font *font, int glyph, float scale, int *width, int *height, int *xoff, int *yoff);

#endif
