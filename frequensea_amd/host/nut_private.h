/* nut_private.h -- helpers shared inside libfsea_nrf.so, not part of the nut.h interface. */
#ifndef FSEA_NUT_PRIVATE_H
#define FSEA_NUT_PRIVATE_H

#include "nut.h"

/* F64 buffer (length x channels) whose payload is NOT zero-filled: for callers that overwrite every
 * element at once (nrf_fft_get_buffer linearising its ring). */
nut_buffer *nut_private_new_f64_unfilled(int n_elements, int n_channels);

#endif
