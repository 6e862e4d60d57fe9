/*
 * nut.h -- typed sample buffers: the boundary type of the frequensea FFT path.
 *
 * Re-statement of the reference's buffer utility so that code written against
 * it links unchanged against libfsea_nrf.so.  Interface being replaced:
 * /root/reference/src/nut.h:14-45 (types + prototypes), semantics from
 * /root/reference/src/nut.c:25-190.  Type tags, struct layout and every
 * function signature are identical; the Lua bindings depend on the field
 * names length / channels / size_bytes (src/main.cpp:121-137).
 */
#ifndef NUT_H
#define NUT_H

#include <stdint.h>

/* Sleep the calling thread (src/nut.c:18-23). */
void nut_sleep_milliseconds(int millis);

/* Element type tags; the numeric values are visible to Lua as
 * NUT_BUFFER_U8 = 1 and NUT_BUFFER_F64 = 2 (src/main.cpp:1181-1190). */
typedef enum {
    NUT_BUFFER_U8 = 1,
    NUT_BUFFER_F64
} nut_buffer_type;

typedef union nut_buffer_data {
    uint8_t *u8;
    double *f64;
} nut_buffer_data;

/* length = elements per channel, channels = interleaved channels,
 * size_bytes = length * channels * element size.  `data` is plain malloc
 * memory owned by the buffer (nut_buffer_free() calls free() on it). */
typedef struct {
    nut_buffer_type type;
    int length;
    int channels;
    int size_bytes;
    nut_buffer_data data;
} nut_buffer;

/* Allocate zero-filled, then copy `data` if it is not NULL (src/nut.c:25-49). */
nut_buffer *nut_buffer_new_u8(int length, int channels, const uint8_t *data);
nut_buffer *nut_buffer_new_f64(int length, int channels, const double *data);
/* Deep copy (src/nut.c:51-58). */
nut_buffer *nut_buffer_copy(nut_buffer *buffer);
/* First round(length * clamp(percentage, 0, 1)) elements (src/nut.c:60-69). */
nut_buffer *nut_buffer_reduce(nut_buffer *buffer, double percentage);
/* `length` elements from element `offset`; length < 0 = to the end
 * (src/nut.c:71-82; the offset is in elements of data, not in frames). */
nut_buffer *nut_buffer_clip(nut_buffer *buffer, int offset, int length);
/* Copy src's payload over dst's; types and sizes must match (src/nut.c:84-94). */
void nut_buffer_set_data(nut_buffer *dst, nut_buffer *src);
/* Grow dst by src's payload; types must match (src/nut.c:96-119). */
void nut_buffer_append(nut_buffer *dst, nut_buffer *src);
/* Element access with the u8 <-> f64 convention f = u / 256.0, u = f * 256.0
 * truncated (src/nut.c:121-151). */
uint8_t nut_buffer_get_u8(nut_buffer *buffer, int offset);
double nut_buffer_get_f64(nut_buffer *buffer, int offset);
void nut_buffer_set_u8(nut_buffer *buffer, int offset, uint8_t value);
void nut_buffer_set_f64(nut_buffer *buffer, int offset, double value);
/* New buffer of the other (or the same) element type (src/nut.c:153-171). */
nut_buffer *nut_buffer_convert(nut_buffer *buffer, nut_buffer_type new_type);
/* Raw dump of the payload to a file (src/nut.c:173-181). */
void nut_buffer_save(nut_buffer *buffer, const char *fname);
void nut_buffer_free(nut_buffer *buffer);

#endif /* NUT_H */
