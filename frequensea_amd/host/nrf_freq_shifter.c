/*
 * nrf_freq_shifter.c -- frequensea's frequency-shifter block (include/nrf.h), the caller that
 * lua/fft-shifted.lua:52-55 puts in front of nrf_fft_process.
 *
 * Reference behaviour restated (paths under /root/reference): src/nrf.c:817-870.  Host
 * arithmetic in double, like the reference: the block's output is a host F64 nut_buffer that
 * the caller owns, so there is nothing to keep on the device (the batched, device-resident
 * form of the same operation is fsea_exec_u8_shifted_device, include/fsea.h).
 * Differences: the output buffer grows if a later input is longer than the first one (the
 * reference would write past it), get_buffer before any process returns NULL instead of
 * dereferencing NULL, and nrf_freq_shifter_free releases the output buffer (leaked upstream).
 */
#include <assert.h>
#include <math.h>
#include <stdlib.h>

#include "nrf.h"

static const double TWO_PI = 6.28318530717958647692;

nrf_freq_shifter *nrf_freq_shifter_new(int freq_offset, int sample_rate) {
    nrf_freq_shifter *shifter = (nrf_freq_shifter *)calloc(1, sizeof(nrf_freq_shifter));
    if (shifter == NULL) return NULL;
    nrf_block_init(&shifter->block, NRF_BLOCK_GENERIC, (nrf_block_process_fn)nrf_freq_shifter_process,
                   (nrf_block_result_fn)nrf_freq_shifter_get_buffer);
    shifter->freq_offset = freq_offset;
    shifter->sample_rate = sample_rate;
    shifter->cosine = 1.0;
    shifter->sine = 0.0;
    return shifter;
}

/* one step of the phase recurrence: (c, s) <- (c, s) rotated by (dc, ds) */
static void advance(double *c, double *s, double dc, double ds) {
    const double next_s = *c * ds + *s * dc;
    const double next_c = *c * dc - *s * ds;
    *s = next_s;
    *c = next_c;
}

void nrf_freq_shifter_process_samples(nrf_freq_shifter *shifter, double *samples_i, double *samples_q, int length) {
    const double step = TWO_PI * shifter->freq_offset / (double)shifter->sample_rate;
    const double dc = cos(step), ds = sin(step);
    double c = shifter->cosine, s = shifter->sine;
    for (int k = 0; k < length; k++) {
        const double vi = samples_i[k], vq = samples_q[k];
        samples_i[k] = vi * c - vq * s;
        samples_q[k] = vi * s + vq * c;
        advance(&c, &s, dc, ds);
    }
    shifter->cosine = c;
    shifter->sine = s;
}

void nrf_freq_shifter_process(nrf_freq_shifter *shifter, nut_buffer *buffer) {
    assert(buffer->channels == 2);
    const int values = buffer->length * buffer->channels;
    if (shifter->buffer != NULL && shifter->buffer->length < values) {
        nut_buffer_free(shifter->buffer);
        shifter->buffer = NULL;
    }
    if (shifter->buffer == NULL) shifter->buffer = nut_buffer_new_f64(values, 2, NULL);
    const double step = TWO_PI * shifter->freq_offset / (double)shifter->sample_rate;
    const double dc = cos(step), ds = sin(step);
    double c = shifter->cosine, s = shifter->sine;
    double *out = shifter->buffer->data.f64;
    for (int k = 0; k < values; k += 2) {
        const double vi = nut_buffer_get_f64(buffer, k);
        const double vq = nut_buffer_get_f64(buffer, k + 1);
        out[k] = vi * c - vq * s + 0.5;
        out[k + 1] = vi * s + vq * c + 0.5;
        advance(&c, &s, dc, ds);
    }
    shifter->cosine = c;
    shifter->sine = s;
}

nut_buffer *nrf_freq_shifter_get_buffer(nrf_freq_shifter *shifter) {
    return shifter->buffer ? nut_buffer_copy(shifter->buffer) : NULL;
}

void nrf_freq_shifter_free(nrf_freq_shifter *shifter) {
    if (shifter == NULL) return;
    if (shifter->buffer != NULL) nut_buffer_free(shifter->buffer);
    free(shifter);
}
