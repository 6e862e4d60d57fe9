/* Test infrastructure: compiles the reference's vendored third-party rasteriser AS IT LIES under
 * /root/reference/externals/stb (stb_truetype.h v1.02, what c/fft-stitch.c:11-12 and c/add-markers.c:11-12
 * include) into oracle/_ref/libstbtt_ref.so, so that tests can pin frequensea_amd/host/ntt_font.c -- glyph
 * indices, metrics, bitmap boxes, coverage -- against the rasteriser the reference's labels come from.
 * Nothing of the header is copied here; oracle/Makefile adds -I$(REF). */
#define STB_TRUETYPE_IMPLEMENTATION
#include "externals/stb/stb_truetype.h"
