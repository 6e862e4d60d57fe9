/*
 * fsea_tune.h -- tuning and measurement additions of libfsea_hip_tune.so.
 *
 * libfsea_hip_tune.so is a superset build of libfsea_hip.so (same sources, -DFSEA_TUNE
 * -DFSEA_TRACE=1): everything include/fsea.h declares, plus the named kernel variants and the
 * measurement-only ablations (frequensea_amd/csrc/fsea_configs_tune.h), per-workgroup time stamps,
 * and a back-to-back timing helper.  Ablation variants ("abl_*") produce wrong spectra by design.
 * Nothing in the product path (libfsea_nrf.so, the tools, bench.py) links or loads this library;
 * scripts/tune.py, scripts/wg_trace.py and friends do.
 */
#ifndef FSEA_TUNE_H
#define FSEA_TUNE_H

#include "fsea.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Same as fsea_plan_create but selects a named kernel variant of the size ("" = the product
 * configuration).  Variants carry the u8 MAG and run-time-mode kernels only.  Unknown variants
 * fail with FSEA_EINVAL. */
int fsea_plan_create_variant(fsea_plan **plan, int fft_size, int hop, int mode, int device,
                             const char *variant);

/* Runs fsea_exec_u8_device `reps` times back to back on `stream` between two HIP events recorded
 * on that same stream and returns the average milliseconds per launch. */
int fsea_time_exec_u8_device(fsea_plan *plan, const void *d_iq, size_t n_frames,
                             int flip, void *d_out, void *stream, int reps,
                             float *avg_ms);

/* The same over n_sets independent buffer sets used in rotation (launch i uses set i % n_sets): with
 * more than 256 MiB in flight no launch is served from the Infinity Cache -- the streaming regime. */
int fsea_time_exec_u8_rotating(fsea_plan *plan, void *const *d_iq, void *const *d_out, int n_sets,
                               size_t n_frames, int flip, void *stream, int reps, float *avg_ms);

/* A plain streaming kernel with the headline kernel's byte mix (every 16 bytes read, 32 written, nt both ways), `reps`
 * launches back to back over n_sets buffer sets in rotation between two HIP events on `stream`: the rate the memory
 * system gives a kernel that does nothing else.  in_bytes per set (a multiple of 16); each d_out holds 2 * in_bytes. */
int fsea_tune_stream_1to2(void *const *d_in, void *const *d_out, int n_sets, size_t in_bytes, int device, void *stream,
                          int reps, float *avg_ms);

/* When the environment variable FSEA_TRACE is set at plan creation, every launch records per
 * workgroup {wall-clock start, end (100 MHz ticks), shader-clock start, end, HW_ID, XCC_ID,
 * prologue done, first pass 0 done, end of iteration 0..23}; this copies the [n_workgroups][32]
 * words of the last launch (scripts/wg_trace.py). */
int fsea_plan_read_trace(fsea_plan *plan, unsigned long long *out, unsigned n_workgroups);

#ifdef __cplusplus
}
#endif
#endif
