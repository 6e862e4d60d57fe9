/*
 * nut.c -- typed sample buffers (include/nut.h).
 *
 * Behaviour follows /root/reference/src/nut.c:25-190 exactly (zero-filled
 * allocation, optional initial copy, u8 <-> f64 as /256.0 and *256.0 with C
 * truncation); the code is organised around one allocator keyed on the element
 * type instead of per-type twins.
 */
#define _POSIX_C_SOURCE 200809L
#include "nut.h"

#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

void nut_sleep_milliseconds(int millis) {
    struct timespec ts;
    ts.tv_sec = millis / 1000;
    ts.tv_nsec = (long)(millis % 1000) * 1000000L;
    nanosleep(&ts, NULL);
}

static size_t elem_size(nut_buffer_type type) {
    return type == NUT_BUFFER_U8 ? sizeof(uint8_t) : sizeof(double);
}

static void *payload(const nut_buffer *b) {
    return b->type == NUT_BUFFER_U8 ? (void *)b->data.u8 : (void *)b->data.f64;
}

static void set_payload(nut_buffer *b, void *p) {
    if (b->type == NUT_BUFFER_U8) {
        b->data.u8 = (uint8_t *)p;
    } else {
        b->data.f64 = (double *)p;
    }
}

/* Zero-filled buffer of `length` x `channels` elements; `init` (may be NULL)
 * supplies the initial contents. */
static nut_buffer *nut_alloc(nut_buffer_type type, int length, int channels, const void *init) {
    nut_buffer *b = (nut_buffer *)calloc(1, sizeof(nut_buffer));
    if (b == NULL) {
        fprintf(stderr, "nut_buffer: out of memory\n");
        exit(EXIT_FAILURE);
    }
    b->type = type;
    b->length = length;
    b->channels = channels;
    b->size_bytes = (int)((size_t)length * (size_t)channels * elem_size(type));
    void *p = calloc(b->size_bytes > 0 ? (size_t)b->size_bytes : 1, 1);
    if (p == NULL) {
        fprintf(stderr, "nut_buffer: out of memory (%d bytes)\n", b->size_bytes);
        exit(EXIT_FAILURE);
    }
    if (init != NULL && b->size_bytes > 0) memcpy(p, init, (size_t)b->size_bytes);
    set_payload(b, p);
    return b;
}

nut_buffer *nut_buffer_new_u8(int length, int channels, const uint8_t *data) {
    return nut_alloc(NUT_BUFFER_U8, length, channels, data);
}

nut_buffer *nut_buffer_new_f64(int length, int channels, const double *data) {
    return nut_alloc(NUT_BUFFER_F64, length, channels, data);
}

nut_buffer *nut_buffer_copy(nut_buffer *buffer) {
    assert(buffer != NULL);
    return nut_alloc(buffer->type, buffer->length, buffer->channels, payload(buffer));
}

nut_buffer *nut_buffer_reduce(nut_buffer *buffer, double percentage) {
    assert(buffer != NULL);
    if (percentage < 0.0) percentage = 0.0;
    if (percentage > 1.0) percentage = 1.0;
    int keep = (int)round(buffer->length * percentage);
    return nut_alloc(buffer->type, keep, buffer->channels, payload(buffer));
}

nut_buffer *nut_buffer_clip(nut_buffer *buffer, int offset, int length) {
    assert(buffer != NULL);
    assert((length < 0) || ((buffer->length - offset) >= length));
    int keep = length;
    if (keep < 0 || keep > buffer->length - offset) keep = buffer->length - offset;
    /* the reference offsets the data pointer by `offset` elements, not frames */
    const uint8_t *from = (const uint8_t *)payload(buffer) + (size_t)offset * elem_size(buffer->type);
    return nut_alloc(buffer->type, keep, buffer->channels, from);
}

void nut_buffer_set_data(nut_buffer *dst, nut_buffer *src) {
    assert(dst != NULL && src != NULL);
    assert(dst->type == src->type);
    assert(dst->size_bytes == src->size_bytes);
    memcpy(payload(dst), payload(src), (size_t)dst->size_bytes);
}

void nut_buffer_append(nut_buffer *dst, nut_buffer *src) {
    assert(dst != NULL && src != NULL);
    assert(dst->type == src->type);
    size_t es = elem_size(dst->type);
    size_t dst_elems = (size_t)dst->length * (size_t)dst->channels;
    size_t src_elems = (size_t)src->length * (size_t)src->channels;
    uint8_t *grown = (uint8_t *)calloc(dst_elems + src_elems ? dst_elems + src_elems : 1, es);
    if (grown == NULL) {
        fprintf(stderr, "nut_buffer_append: out of memory\n");
        exit(EXIT_FAILURE);
    }
    memcpy(grown, payload(dst), (size_t)dst->size_bytes);
    memcpy(grown + dst_elems * es, payload(src), (size_t)src->size_bytes);
    free(payload(dst));
    set_payload(dst, grown);
    dst->size_bytes = (int)((dst_elems + src_elems) * es);
    dst->length += src->length;
}

uint8_t nut_buffer_get_u8(nut_buffer *buffer, int offset) {
    if (buffer->type == NUT_BUFFER_U8) return buffer->data.u8[offset];
    return (uint8_t)(buffer->data.f64[offset] * 256.0);
}

double nut_buffer_get_f64(nut_buffer *buffer, int offset) {
    if (buffer->type == NUT_BUFFER_F64) return buffer->data.f64[offset];
    return buffer->data.u8[offset] / 256.0;
}

void nut_buffer_set_u8(nut_buffer *buffer, int offset, uint8_t value) {
    if (buffer->type == NUT_BUFFER_U8) {
        buffer->data.u8[offset] = value;
    } else {
        buffer->data.f64[offset] = value / 256.0;
    }
}

void nut_buffer_set_f64(nut_buffer *buffer, int offset, double value) {
    if (buffer->type == NUT_BUFFER_F64) {
        buffer->data.f64[offset] = value;
    } else {
        buffer->data.u8[offset] = (uint8_t)(value * 256.0);
    }
}

nut_buffer *nut_buffer_convert(nut_buffer *buffer, nut_buffer_type new_type) {
    assert(buffer != NULL);
    nut_buffer *out = nut_alloc(new_type, buffer->length, buffer->channels, NULL);
    int count = buffer->length * buffer->channels;
    for (int i = 0; i < count; i++) {
        if (new_type == NUT_BUFFER_U8) {
            out->data.u8[i] = nut_buffer_get_u8(buffer, i);
        } else {
            out->data.f64[i] = nut_buffer_get_f64(buffer, i);
        }
    }
    return out;
}

void nut_buffer_save(nut_buffer *buffer, const char *fname) {
    assert(buffer != NULL);
    FILE *fp = fopen(fname, "wb");
    if (fp == NULL) return;
    fwrite(payload(buffer), (size_t)buffer->size_bytes, 1, fp);
    fclose(fp);
    printf("Written %s.\n", fname);
}

void nut_buffer_free(nut_buffer *buffer) {
    if (buffer == NULL) return;
    free(payload(buffer));
    free(buffer);
}
