/*
 * nrf_device.c -- block graph + the file-replay ("dummy") sample source of
 * frequensea (include/nrf.h).  The SDR hardware drivers are out of scope: no
 * radio exists on a GPU node, so nrf_device_new always takes the path the
 * reference falls back to when neither RTL-SDR nor HackRF opens.
 *
 * Reference behaviour restated (paths under /root/reference):
 *   nrf_block_*                    src/nrf.c:24-50
 *   sample-block ingest + flip     src/nrf.c:95-131
 *   replay loop / stepping         src/nrf.c:153-170, 341-350
 *   file load                      src/nrf.c:256-284
 *   nrf_device_new*                src/nrf.c:289-322
 *   nrf_device_get_samples_buffer  src/nrf.c:352-357
 *   nrf_device_free                src/nrf.c:426-444
 */
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nrf.h"

/* ---- block graph --------------------------------------------------------- */

void nrf_block_init(nrf_block *block, nrf_block_type type, nrf_block_process_fn process_fn,
                    nrf_block_result_fn result_fn) {
    assert(block->n_outputs == 0);
    block->type = type;
    block->process_fn = process_fn;
    block->result_fn = result_fn;
}

void nrf_block_connect(nrf_block *input, nrf_block *output) {
    assert(input->n_outputs < NRF_BLOCK_MAX_OUTPUTS);
    input->outputs[input->n_outputs++] = output;
}

void nrf_block_process(nrf_block *block, nut_buffer *buffer) {
    if (block->process_fn != NULL) block->process_fn(block, buffer);
    if (block->n_outputs == 0) return;
    nut_buffer *result = block->result_fn(block);
    for (int i = 0; i < block->n_outputs; i++) {
        nrf_block_process((nrf_block *)block->outputs[i], result);
    }
    nut_buffer_free(result);
}

/* ---- replay source -------------------------------------------------------- */

static const int REPLAY_SAMPLE_RATE = 5000000; /* src/nrf.c:254 */

/* One 262144-byte block of raw int8 IQ -> device->samples in offset binary:
 * u = (b + 128) % 256 == b ^ 0x80, eight bytes per step. */
static void ingest_block(nrf_device *device, const uint8_t *block) {
    if (!device->receiving) return;
    pthread_mutex_lock(&device->data_mutex);
    for (size_t i = 0; i < NRF_BUFFER_SIZE_BYTES; i += 8) {
        uint64_t w;
        memcpy(&w, block + i, 8);
        w ^= UINT64_C(0x8080808080808080);
        memcpy(device->samples + i, &w, 8);
    }
    pthread_mutex_unlock(&device->data_mutex);
    if (device->decode_cb_fn != NULL) device->decode_cb_fn(device, device->decode_cb_ctx);
    if (!device->receiving) return;
    nrf_block_process(&device->block, NULL);
}

static void *replay_loop(void *arg) {
    nrf_device *device = (nrf_device *)arg;
    while (device->receiving) {
        ingest_block(device, device->receive_buffer + (size_t)device->dummy_block_index * NRF_BUFFER_SIZE_BYTES);
        if (!device->paused) nrf_device_step(device);
        nut_sleep_milliseconds(1000 / 60);
    }
    return NULL;
}

static void replay_start(nrf_device *device, const char *data_file) {
    device->device_type = NRF_DEVICE_DUMMY;
    device->sample_rate = REPLAY_SAMPLE_RATE;
    fprintf(stderr, "WARN nrf_device_new: no SDR device in this build. Replaying data file %s\n",
            data_file ? data_file : "(none)");
    FILE *fp = data_file ? fopen(data_file, "rb") : NULL;
    long size = 0;
    if (fp != NULL) {
        fseek(fp, 0L, SEEK_END);
        size = ftell(fp);
        rewind(fp);
    }
    if (fp != NULL && size >= NRF_BUFFER_SIZE_BYTES) {
        device->receive_buffer = (uint8_t *)calloc((size_t)size, 1);
        device->dummy_block_length = (int)(size / NRF_BUFFER_SIZE_BYTES);
        if (fread(device->receive_buffer, (size_t)size, 1, fp) != 1) {
            fprintf(stderr, "WARN nrf_device_new: short read on %s\n", data_file);
        }
    } else {
        /* missing file (as the reference) or a file shorter than one block (the
         * reference would index past its buffer: src/nrf.c:268) -> one zero block */
        if (data_file != NULL) fprintf(stderr, "WARN nrf_device_new: Couldn't use %s. Using empty buffer.\n", data_file);
        device->receive_buffer = (uint8_t *)calloc(NRF_BUFFER_SIZE_BYTES, 1);
        device->dummy_block_length = 1;
    }
    if (fp != NULL) fclose(fp);
    device->dummy_block_index = 0;
    device->receiving = 1;
    pthread_create(&device->receive_thread, NULL, replay_loop, device);
}

nrf_device *nrf_device_new(double freq_mhz, const char *data_file) {
    nrf_device_config config;
    memset(&config, 0, sizeof(config));
    config.freq_mhz = freq_mhz;
    config.data_file = data_file;
    return nrf_device_new_with_config(config);
}

nrf_device *nrf_device_new_with_config(const nrf_device_config config) {
    nrf_device *device = (nrf_device *)calloc(1, sizeof(nrf_device));
    if (device == NULL) {
        fprintf(stderr, "ERROR nrf_device_new: out of memory\n");
        exit(EXIT_FAILURE);
    }
    nrf_block_init(&device->block, NRF_BLOCK_SOURCE, NULL, (nrf_block_result_fn)nrf_device_get_samples_buffer);
    pthread_mutex_init(&device->data_mutex, NULL);
    replay_start(device, config.data_file);
    if (config.sample_rate > 0) device->sample_rate = config.sample_rate;
    return device;
}

double nrf_device_set_frequency(nrf_device *device, double freq_mhz) {
    (void)device; /* a file has no tuner; the dummy device accepts any value (src/nrf.c:86-94) */
    return freq_mhz;
}

void nrf_device_set_decode_handler(nrf_device *device, nrf_device_decode_cb_fn fn, void *ctx) {
    device->decode_cb_fn = fn;
    device->decode_cb_ctx = ctx;
}

void nrf_device_set_paused(nrf_device *device, int paused) { device->paused = paused; }

void nrf_device_step(nrf_device *device) {
    int next = device->dummy_block_index + 1;
    device->dummy_block_index = next >= device->dummy_block_length ? 0 : next;
}

nut_buffer *nrf_device_get_samples_buffer(nrf_device *device) {
    pthread_mutex_lock(&device->data_mutex);
    nut_buffer *snapshot = nut_buffer_new_u8(NRF_SAMPLES_LENGTH, 2, device->samples);
    pthread_mutex_unlock(&device->data_mutex);
    return snapshot;
}

void nrf_device_free(nrf_device *device) {
    if (device == NULL) return;
    device->receiving = 0;
    pthread_join(device->receive_thread, NULL);
    pthread_mutex_destroy(&device->data_mutex);
    free(device->receive_buffer);
    free(device);
}
