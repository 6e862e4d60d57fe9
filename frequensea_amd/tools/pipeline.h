/*
 * pipeline.h -- the host side of the sweep tools as three overlapped stages (round 3).
 *
 * The reference's tools are streaming consumers: one transfer comes in, one spectrum row goes out, inside the
 * radio's receive callback (c/fft-batch.c:54-102).  With captures on disk and the transform on a GPU, the time of a
 * sweep is file reading and PNG encoding, not the FFT; so while capture k is on the GPU a reader thread is already
 * loading capture k + 1 and a writer thread is still encoding the PNG of capture k - 1.  Two buffers per stage
 * boundary, handed over under a mutex: a slot is either free (the producer may fill it) or full (the consumer may
 * take it).  Buffers are page-locked (fsea_host_alloc) so that the copies to and from the device run at link rate.
 */
#ifndef FSEA_TOOLS_PIPELINE_H
#define FSEA_TOOLS_PIPELINE_H

#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#include "easypng.h"

typedef struct {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int full[2];
} slot_ring;

static inline double stage_clock(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static inline void ring_init(slot_ring *r) {
    pthread_mutex_init(&r->mu, NULL);
    pthread_cond_init(&r->cv, NULL);
    r->full[0] = r->full[1] = 0;
}
static inline void ring_wait(slot_ring *r, int slot, int want_full) {
    pthread_mutex_lock(&r->mu);
    while (r->full[slot] != want_full) pthread_cond_wait(&r->cv, &r->mu);
    pthread_mutex_unlock(&r->mu);
}
/* ring_wait with a deadline: 0 = the slot got there, -1 = it did not within `seconds` (the fatal paths: a process that is
 * about to exit must not wait for ever on a writer whose owner is gone) */
static inline struct timespec ring_deadline(double seconds) {
    struct timespec until;
    clock_gettime(CLOCK_REALTIME, &until);
    if (seconds < 0) seconds = 0;
    until.tv_sec += (time_t)seconds;
    until.tv_nsec += (long)((seconds - (double)(time_t)seconds) * 1e9);
    if (until.tv_nsec >= 1000000000L) {
        until.tv_sec += 1;
        until.tv_nsec -= 1000000000L;
    }
    return until;
}
/* the deadline is ABSOLUTE (CLOCK_REALTIME): one deadline can bound a whole sequence of waits */
static inline int ring_wait_until(slot_ring *r, int slot, int want_full, const struct timespec *until) {
    int rc = 0;
    pthread_mutex_lock(&r->mu);
    while (r->full[slot] != want_full && rc == 0) rc = pthread_cond_timedwait(&r->cv, &r->mu, until);
    const int ok = r->full[slot] == want_full;
    pthread_mutex_unlock(&r->mu);
    return ok ? 0 : -1;
}
static inline int ring_wait_for(slot_ring *r, int slot, int want_full, double seconds) {
    const struct timespec until = ring_deadline(seconds);
    return ring_wait_until(r, slot, want_full, &until);
}
static inline void ring_set(slot_ring *r, int slot, int full) {
    pthread_mutex_lock(&r->mu);
    r->full[slot] = full;
    pthread_cond_broadcast(&r->cv);
    pthread_mutex_unlock(&r->mu);
}

/* ---- reader: item i is loaded into buffer i % 2 by `load` (returns 0, or non-zero = this item failed) ---- */
typedef int (*capture_loader)(void *ctx, int item, uint8_t *buffer, int *rows_out);
typedef struct {
    slot_ring ring;
    pthread_t thread;
    int n_items;
    capture_loader load;
    void *ctx;
    uint8_t *buffer[2];
    int rc[2], rows[2];
    double busy_s; /* seconds spent loading (--timing) */
} capture_reader;

static inline void *capture_reader_main(void *p) {
    capture_reader *r = (capture_reader *)p;
    for (int i = 0; i < r->n_items; i++) {
        const int s = i & 1;
        ring_wait(&r->ring, s, 0);
        r->rows[s] = 0;
        const double t0 = stage_clock();
        r->rc[s] = r->load(r->ctx, i, r->buffer[s], &r->rows[s]);
        r->busy_s += stage_clock() - t0;
        ring_set(&r->ring, s, 1);
    }
    return NULL;
}
static inline int capture_reader_start(capture_reader *r, int n_items, capture_loader load, void *ctx, uint8_t *b0, uint8_t *b1) {
    ring_init(&r->ring);
    r->busy_s = 0.0;
    r->n_items = n_items;
    r->load = load;
    r->ctx = ctx;
    r->buffer[0] = b0;
    r->buffer[1] = b1;
    return pthread_create(&r->thread, NULL, capture_reader_main, r);
}
/* item i (in order): waits until it is loaded; *buffer stays valid until capture_reader_release(i) */
static inline int capture_reader_take(capture_reader *r, int item, uint8_t **buffer, int *rows) {
    const int s = item & 1;
    ring_wait(&r->ring, s, 1);
    *buffer = r->buffer[s];
    *rows = r->rows[s];
    return r->rc[s];
}
static inline void capture_reader_release(capture_reader *r, int item) { ring_set(&r->ring, item & 1, 0); }
static inline void capture_reader_join(capture_reader *r) { pthread_join(r->thread, NULL); }

/* ---- writer: PNGs are encoded and written from pixel buffer k % 2 of job k ---- */
typedef struct {
    slot_ring ring;
    pthread_t thread;
    uint8_t *pixels[2];
    char name[2][600];
    int width[2], height[2];
    int stop, jobs, failed;
    double busy_s; /* seconds spent encoding and writing (--timing) */
} png_writer;

static inline void *png_writer_main(void *p) {
    png_writer *w = (png_writer *)p;
    for (int k = 0;; k++) {
        const int s = k & 1;
        ring_wait(&w->ring, s, 1);
        if (w->height[s] < 0) break; /* the stop marker */
        const double t0 = stage_clock();
        if (write_gray_png(w->name[s], w->width[s], w->height[s], w->pixels[s]) != 0) w->failed = 1;
        w->busy_s += stage_clock() - t0;
        ring_set(&w->ring, s, 0);
    }
    return NULL;
}
static inline int png_writer_start(png_writer *w, uint8_t *p0, uint8_t *p1) {
    ring_init(&w->ring);
    w->pixels[0] = p0;
    w->pixels[1] = p1;
    w->stop = w->jobs = w->failed = 0;
    w->busy_s = 0.0;
    return pthread_create(&w->thread, NULL, png_writer_main, w);
}
/* the pixel buffer of the next job, free to be filled (waits for the job that used it two jobs ago) */
static inline uint8_t *png_writer_acquire(png_writer *w) {
    const int s = w->jobs & 1;
    ring_wait(&w->ring, s, 0);
    return w->pixels[s];
}
static inline void png_writer_submit(png_writer *w, const char *name, int width, int height) {
    const int s = w->jobs & 1;
    snprintf(w->name[s], sizeof(w->name[s]), "%s", name);
    w->width[s] = width;
    w->height[s] = height;
    w->jobs++;
    ring_set(&w->ring, s, 1);
}
/* The same two calls by job number, for a caller that fills the buffer of job k + 1 before it submits job k (two batches
 * in flight on two streams): job k uses pixel buffer k % 2; submissions must still be made in job order. */
static inline uint8_t *png_writer_acquire_job(png_writer *w, int job) {
    ring_wait(&w->ring, job & 1, 0);
    return w->pixels[job & 1];
}
static inline void png_writer_submit_job(png_writer *w, int job, const char *name, int width, int height) {
    const int s = job & 1;
    snprintf(w->name[s], sizeof(w->name[s]), "%s", name);
    w->width[s] = width;
    w->height[s] = height;
    w->jobs = job + 1;
    ring_set(&w->ring, s, 1);
}
/* Waits until no submitted PNG is pending (both buffers empty); callable from any thread -- the fatal paths use it so that
 * an error in one capture does not truncate the PNG of an earlier, successfully processed one. */
static inline void png_writer_drain(png_writer *w) {
    ring_wait(&w->ring, 0, 0);
    ring_wait(&w->ring, 1, 0);
}
/* the same with a bound on the wait; -1 = a PNG was still pending after `seconds` */
static inline int png_writer_drain_until(png_writer *w, const struct timespec *until) {
    const int a = ring_wait_until(&w->ring, 0, 0, until);
    const int b = ring_wait_until(&w->ring, 1, 0, until);
    return (a == 0 && b == 0) ? 0 : -1;
}
static inline int png_writer_drain_for(png_writer *w, double seconds) {
    const struct timespec until = ring_deadline(seconds);   /* one deadline for both slots */
    return png_writer_drain_until(w, &until);
}
/* waits for every submitted PNG; returns non-zero if one of them could not be written */
static inline int png_writer_finish(png_writer *w) {
    const int s = w->jobs & 1;
    ring_wait(&w->ring, s, 0);
    w->height[s] = -1;
    ring_set(&w->ring, s, 1);
    pthread_join(w->thread, NULL);
    return w->failed;
}

#endif
