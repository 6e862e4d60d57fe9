/*
 * easypng.h -- 8-bit grayscale PNG in and out: the on-disk format of the FFT sweep.
 *
 * Interface being replaced (paths under /root/reference):
 *   c/easypng.h:6-53        write_gray_png(fname, width, height, buffer) via libpng
 *   c/fft-stitch.c:172-183  stbi_load(file_name, &width, &height, &n, 1) via stb_image
 * Same file format (8-bit gray, non-interlaced); implemented directly on zlib (this image has
 * libpng's runtime but not its headers, and stb_image is the reference's vendored code).
 */
#ifndef FSEA_EASYPNG_H
#define FSEA_EASYPNG_H

#include <stddef.h>
#include <stdint.h>

/* Writes buffer[height][width] as an 8-bit gray PNG and prints "Written <fname>." like the
 * reference.  Returns 0, or -1 (after printing an ERROR line) when the file cannot be written. */
int write_gray_png(const char *fname, int width, int height, const uint8_t *buffer);
/* The same with the compressed data cut into IDAT chunks of at most `idat_max` bytes (libpng, which the reference links,
 * writes 8192-byte chunks; write_gray_png uses 2^30, the most a 31-bit chunk length safely holds: the reference's own
 * stitched image, 154112 x 11811 pixels (c/fft-stitch.c:16-27), is 1.8 GB of scanlines). */
int write_gray_png_chunked(const char *fname, int width, int height, const uint8_t *buffer, size_t idat_max);

/* Reads an 8-bit gray (or gray+alpha / RGB / RGBA, converted to gray the way stb_image does
 * for req_comp = 1) non-interlaced PNG.  Returns malloc'd pixels [height][width] or NULL. */
uint8_t *read_gray_png(const char *fname, int *width, int *height);

#endif
