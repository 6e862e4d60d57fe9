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
#include "nut_private.h"

#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

void nut_sleep_milliseconds(int duration_ms) {
    struct timespec ts;
    ts.tv_sec = duration_ms / 1000;
    ts.tv_nsec = (long)(duration_ms % 1000) * 1000000L;
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

/* Buffer of `length` x `channels` elements: a copy of `init`, or zero-filled when init is NULL.
 * zeroed = 0 skips the zero fill for callers that overwrite every element themselves. */
static nut_buffer *nut_alloc_mode(nut_buffer_type type, int length, int channels, const void *init, int zeroed) {
    nut_buffer *b = (nut_buffer *)calloc(1, sizeof(nut_buffer));
    if (b == NULL) {
        fprintf(stderr, "nut_buffer: out of memory\n");
        exit(EXIT_FAILURE);
    }
    b->type = type;
    b->length = length;
    b->channels = channels;
    b->size_bytes = (int)((size_t)length * (size_t)channels * elem_size(type));
    const size_t bytes = b->size_bytes > 0 ? (size_t)b->size_bytes : 1;
    /* a recycled heap chunk would be memset by calloc only to be overwritten by the copy below */
    void *p = (init != NULL || !zeroed) ? malloc(bytes) : calloc(bytes, 1);
    if (p == NULL) {
        fprintf(stderr, "nut_buffer: out of memory (%d bytes)\n", b->size_bytes);
        exit(EXIT_FAILURE);
    }
    if (init != NULL && b->size_bytes > 0) memcpy(p, init, (size_t)b->size_bytes);
    set_payload(b, p);
    return b;
}

static nut_buffer *nut_alloc(nut_buffer_type type, int length, int channels, const void *init) {
    return nut_alloc_mode(type, length, channels, init, 1);
}

/* Library-internal (nut_private.h): F64 buffer whose contents the caller is about to write in full. */
nut_buffer *nut_private_new_f64_unfilled(int n_elements, int n_channels) {
    return nut_alloc_mode(NUT_BUFFER_F64, n_elements, n_channels, NULL, 0);
}

nut_buffer *nut_buffer_new_u8(int n_elements, int n_channels, const uint8_t *initial) {
    return nut_alloc(NUT_BUFFER_U8, n_elements, n_channels, initial);
}

nut_buffer *nut_buffer_new_f64(int n_elements, int n_channels, const double *initial) {
    return nut_alloc(NUT_BUFFER_F64, n_elements, n_channels, initial);
}

nut_buffer *nut_buffer_copy(nut_buffer *source) {
    assert(source != NULL);
    return nut_alloc(source->type, source->length, source->channels, payload(source));
}

nut_buffer *nut_buffer_reduce(nut_buffer *source, double fraction) {
    assert(source != NULL);
    const double f = fraction < 0.0 ? 0.0 : (fraction > 1.0 ? 1.0 : fraction);
    return nut_alloc(source->type, (int)round(source->length * f), source->channels, payload(source));
}

nut_buffer *nut_buffer_clip(nut_buffer *source, int first, int count) {
    assert(source != NULL);
    const int available = source->length - first;
    assert(count < 0 || available >= count);
    const int keep = (count < 0 || count > available) ? available : count;
    /* the reference advances the data pointer by `first` elements, not frames */
    const uint8_t *from = (const uint8_t *)payload(source) + (size_t)first * elem_size(source->type);
    return nut_alloc(source->type, keep, source->channels, from);
}

void nut_buffer_set_data(nut_buffer *target, nut_buffer *origin) {
    assert(target != NULL && origin != NULL);
    assert(target->type == origin->type && target->size_bytes == origin->size_bytes);
    memcpy(payload(target), payload(origin), (size_t)target->size_bytes);
}

void nut_buffer_append(nut_buffer *target, nut_buffer *origin) {
    assert(target != NULL && origin != NULL);
    assert(target->type == origin->type);
    size_t es = elem_size(target->type);
    size_t target_elems = (size_t)target->length * (size_t)target->channels;
    size_t origin_elems = (size_t)origin->length * (size_t)origin->channels;
    uint8_t *grown = (uint8_t *)calloc(target_elems + origin_elems ? target_elems + origin_elems : 1, es);
    if (grown == NULL) {
        fprintf(stderr, "nut_buffer_append: out of memory\n");
        exit(EXIT_FAILURE);
    }
    memcpy(grown, payload(target), (size_t)target->size_bytes);
    memcpy(grown + target_elems * es, payload(origin), (size_t)origin->size_bytes);
    free(payload(target));
    set_payload(target, grown);
    target->size_bytes = (int)((target_elems + origin_elems) * es);
    target->length += origin->length;
}

uint8_t nut_buffer_get_u8(nut_buffer *source, int element) {
    if (source->type == NUT_BUFFER_U8) return source->data.u8[element];
    return (uint8_t)(source->data.f64[element] * 256.0);
}

double nut_buffer_get_f64(nut_buffer *source, int element) {
    if (source->type == NUT_BUFFER_F64) return source->data.f64[element];
    return source->data.u8[element] / 256.0;
}

void nut_buffer_set_u8(nut_buffer *target, int element, uint8_t sample) {
    if (target->type == NUT_BUFFER_U8) {
        target->data.u8[element] = sample;
    } else {
        target->data.f64[element] = sample / 256.0;
    }
}

void nut_buffer_set_f64(nut_buffer *target, int element, double sample) {
    if (target->type == NUT_BUFFER_F64) {
        target->data.f64[element] = sample;
    } else {
        target->data.u8[element] = (uint8_t)(sample * 256.0);
    }
}

nut_buffer *nut_buffer_convert(nut_buffer *source, nut_buffer_type wanted) {
    assert(source != NULL);
    nut_buffer *converted = nut_alloc(wanted, source->length, source->channels, NULL);
    const int total = source->length * source->channels;
    if (wanted == NUT_BUFFER_U8) {
        for (int i = 0; i < total; i++) converted->data.u8[i] = nut_buffer_get_u8(source, i);
    } else {
        for (int i = 0; i < total; i++) converted->data.f64[i] = nut_buffer_get_f64(source, i);
    }
    return converted;
}

void nut_buffer_save(nut_buffer *source, const char *path) {
    assert(source != NULL);
    FILE *out = fopen(path, "wb");
    if (out == NULL) return;
    fwrite(payload(source), (size_t)source->size_bytes, 1, out);
    fclose(out);
    printf("Written %s.\n", path);
}

void nut_buffer_free(nut_buffer *victim) {
    if (victim == NULL) return;
    free(payload(victim));
    free(victim);
}
