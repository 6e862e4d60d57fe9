/*
 * nut.h -- typed sample buffers: the boundary type of the frequensea FFT path.
 *
 * Every function below keeps the C signature (return type, parameter types and order) and the
 * behaviour of the reference's buffer utility, so code written against that utility -- the Lua
 * wrappers of src/main.cpp:115-179, the nrf blocks -- links against libfsea_nrf.so unchanged.
 * Interface being replaced: /root/reference/src/nut.h:14-45; semantics from
 * /root/reference/src/nut.c:25-190.  The Lua bindings additionally depend on the struct field
 * names length / channels / size_bytes (src/main.cpp:121-137) and on the numeric values of the
 * type tags (src/main.cpp:1181-1190), which is why those are spelled exactly as upstream.
 */
#ifndef NUT_H
#define NUT_H

#include <stdint.h>

/* ---- element types -------------------------------------------------------------------------
 * NUT_BUFFER_U8  = 1   unsigned 8-bit samples (offset-binary IQ as the device delivers it)
 * NUT_BUFFER_F64 = 2   double precision (spectrum history, converted samples)
 * The two views share the convention  f64 = u8 / 256.0  and  u8 = (uint8_t)(f64 * 256.0). */
typedef enum {
    NUT_BUFFER_U8 = 1,
    NUT_BUFFER_F64
} nut_buffer_type;

/* The payload pointer, viewed per element type. */
typedef union nut_buffer_data {
    uint8_t *u8;
    double *f64;
} nut_buffer_data;

/* A buffer owns `size_bytes` bytes of plain malloc memory:
 *   length      elements per channel
 *   channels    interleaved channels (IQ = 2, spectrum history = 1)
 *   size_bytes  length * channels * sizeof(element)
 * Whoever holds the buffer releases it with nut_buffer_free(), which free()s the payload --
 * so the payload must never be pinned / device memory. */
typedef struct {
    nut_buffer_type type;
    int length;
    int channels;
    int size_bytes;
    nut_buffer_data data;
} nut_buffer;

/* ---- creation ------------------------------------------------------------------------------
 * Zero-filled allocation; when `initial` is not NULL its first size_bytes bytes are copied in
 * (src/nut.c:25-49). */
nut_buffer *nut_buffer_new_u8(int n_elements, int n_channels, const uint8_t *initial);
nut_buffer *nut_buffer_new_f64(int n_elements, int n_channels, const double *initial);

/* Deep copy of `source` (src/nut.c:51-58). */
nut_buffer *nut_buffer_copy(nut_buffer *source);

/* New buffer holding the first round(length * fraction) elements, fraction clamped to [0, 1]
 * (src/nut.c:60-69). */
nut_buffer *nut_buffer_reduce(nut_buffer *source, double fraction);

/* New buffer of `count` elements starting `first` elements into the payload; count < 0 means
 * "to the end" (src/nut.c:71-82 -- the offset is applied to the data pointer in elements, not in
 * frames of `channels`). */
nut_buffer *nut_buffer_clip(nut_buffer *source, int first, int count);

/* New buffer of element type `wanted` with every element converted (src/nut.c:153-171). */
nut_buffer *nut_buffer_convert(nut_buffer *source, nut_buffer_type wanted);

/* ---- in-place updates ----------------------------------------------------------------------
 * Overwrite target's payload with origin's; both must have the same type and size
 * (src/nut.c:84-94). */
void nut_buffer_set_data(nut_buffer *target, nut_buffer *origin);

/* Grow target by origin's payload; both must have the same type (src/nut.c:96-119). */
void nut_buffer_append(nut_buffer *target, nut_buffer *origin);

/* ---- element access with implicit conversion (src/nut.c:121-151) --------------------------- */
uint8_t nut_buffer_get_u8(nut_buffer *source, int element);
double nut_buffer_get_f64(nut_buffer *source, int element);
void nut_buffer_set_u8(nut_buffer *target, int element, uint8_t sample);
void nut_buffer_set_f64(nut_buffer *target, int element, double sample);

/* ---- output / lifetime ---------------------------------------------------------------------
 * Raw dump of the payload to `path`; prints "Written <path>." (src/nut.c:173-181). */
void nut_buffer_save(nut_buffer *source, const char *path);
void nut_buffer_free(nut_buffer *victim);

/* Sleep the calling thread for `duration_ms` milliseconds (src/nut.c:18-23). */
void nut_sleep_milliseconds(int duration_ms);

#endif /* NUT_H */
