// fsea_api.hip -- the C ABI of libfsea_hip.so (see include/fsea.h).
//
// Host-side plan management around the kernels of fsea_fft_core.h: twiddle
// tables, persistent-grid sizing, launches, and the two small helper kernels
// (tile max-composite, mean-magnitude reduction).  No CPU compute path exists
// here: if HIP cannot give us a device, plan creation fails.
#include "../../include/fsea.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "fsea_registry.h"
#include "fsea_tables.h"
#include "fsea_internal.h"

using namespace fsea_detail;

extern "C" int fsea_kernels_small(fsea::KernelEntry *out, int cap);
extern "C" int fsea_kernels_alt(fsea::KernelEntry *out, int cap);
extern "C" int fsea_kernels_1024(fsea::KernelEntry *out, int cap);
extern "C" int fsea_kernels_2048(fsea::KernelEntry *out, int cap);
extern "C" int fsea_kernels_4096(fsea::KernelEntry *out, int cap);
extern "C" int fsea_kernels_8192(fsea::KernelEntry *out, int cap);
extern "C" int fsea_kernels_16384(fsea::KernelEntry *out, int cap);
#ifdef FSEA_TUNE  // libfsea_hip_tune.so: the product kernels plus the tuning variants and ablations
#include "../../include/fsea_tune.h"
extern "C" int fsea_kernels_tune_8192a(fsea::KernelEntry *out, int cap);
extern "C" int fsea_kernels_tune_8192b(fsea::KernelEntry *out, int cap);
extern "C" int fsea_kernels_tune_abl(fsea::KernelEntry *out, int cap);
extern "C" int fsea_kernels_tune_mid(fsea::KernelEntry *out, int cap);
extern "C" int fsea_kernels_tune_px(fsea::KernelEntry *out, int cap);
extern "C" int fsea_kernels_tune_big(fsea::KernelEntry *out, int cap);
extern "C" int fsea_kernels_tune_w64(fsea::KernelEntry *out, int cap);
extern "C" int fsea_kernels_tune_lay(fsea::KernelEntry *out, int cap);
extern "C" int fsea_kernels_tune_pw(fsea::KernelEntry *out, int cap);
extern "C" int fsea_kernels_tune_win(fsea::KernelEntry *out, int cap);
#endif

namespace {
thread_local std::string g_last_error = "";
}

namespace fsea_detail {

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

}  // namespace fsea_detail

namespace {

const std::vector<fsea::KernelEntry> &registry() {
    static std::vector<fsea::KernelEntry> all = [] {
        std::vector<fsea::KernelEntry> v;
        fsea::KernelEntry tmp[32];
        int (*lists[])(fsea::KernelEntry *, int) = {fsea_kernels_small, fsea_kernels_alt, fsea_kernels_1024, fsea_kernels_2048,
                                                    fsea_kernels_4096,  fsea_kernels_8192, fsea_kernels_16384,
#ifdef FSEA_TUNE
                                                    fsea_kernels_tune_8192a, fsea_kernels_tune_8192b,
                                                    fsea_kernels_tune_abl, fsea_kernels_tune_px, fsea_kernels_tune_mid, fsea_kernels_tune_big,
                                                    fsea_kernels_tune_w64, fsea_kernels_tune_lay, fsea_kernels_tune_pw, fsea_kernels_tune_win,
#endif
        };
        for (auto fn : lists) {
            int n = fn(tmp, 32);
            for (int i = 0; i < n; ++i) v.push_back(tmp[i]);
        }
        return v;
    }();
    return all;
}

const fsea::KernelEntry *find_entry(int n, const char *variant) {
    for (const auto &e : registry()) {
        if (e.n == n && std::strcmp(e.variant, variant ? variant : "") == 0) return &e;
    }
    return nullptr;
}

}  // namespace

namespace fsea_detail {
size_t mode_elem_bytes(int mode) {
    switch (mode) {
    case FSEA_MODE_DB10_U8:
    case FSEA_MODE_DB5_U8_DCFIX:
        return 1;
    case FSEA_MODE_COMPLEX_F32:
        return 8;
    default:
        return 4;
    }
}

}  // namespace fsea_detail

namespace {

// the windowed twin of a u8 kind (a plan with a taper window; fsea_plan_set_window)
int windowed_kind(int kind) {
    switch (kind) {
    case fsea::K_U8_MAG: return fsea::K_U8_MAG_WIN;
    case fsea::K_U8_DB5: return fsea::K_U8_DB5_WIN;
    case fsea::K_U8_DB10: return fsea::K_U8_DB10_WIN;
    default: return fsea::K_U8_WIN;
    }
}

// which __global__ entry point serves (input kind, epilogue mode, byte convention)
int pick_kind(int in_kind, int mode, int flip) {
    if (in_kind == fsea::IN_F32) return fsea::K_F32;
    if (in_kind == fsea::IN_U8_ROT) return fsea::K_U8_ROT;
    if (flip && mode == FSEA_MODE_MAG_F32) return fsea::K_U8_MAG;
    if (flip && mode == FSEA_MODE_DB5_U8_DCFIX) return fsea::K_U8_DB5;
    if (flip && mode == FSEA_MODE_DB10_U8) return fsea::K_U8_DB10;
    return fsea::K_U8;
}

__global__ void fsea_composite_max_kernel(uint8_t *dst, const uint8_t *src, uint32_t dst_x, uint32_t dst_y,
                                          uint32_t width, uint32_t height, uint32_t dst_stride,
                                          uint32_t src_stride) {
    // one thread per 4 horizontally adjacent pixels; rows by blockIdx.y
    const uint32_t x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    for (uint32_t y = blockIdx.y; y < height; y += gridDim.y) {
        if (x4 >= width) return;
        uint8_t *d = dst + (size_t)(dst_y + y) * dst_stride + dst_x + x4;
        const uint8_t *s = src + (size_t)y * src_stride + x4;
        const bool vec = (x4 + 4 <= width) && ((reinterpret_cast<uintptr_t>(d) & 3) == 0) &&
                         ((reinterpret_cast<uintptr_t>(s) & 3) == 0);
        if (vec) {
            const uint32_t a = *reinterpret_cast<const uint32_t *>(d);
            const uint32_t b = *reinterpret_cast<const uint32_t *>(s);
            uint32_t r = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t ab = (a >> (8 * k)) & 0xff, bb = (b >> (8 * k)) & 0xff;
                r |= (ab > bb ? ab : bb) << (8 * k);
            }
            *reinterpret_cast<uint32_t *>(d) = r;
        } else {
            for (uint32_t k = 0; k < 4 && x4 + k < width; ++k) d[k] = d[k] > s[k] ? d[k] : s[k];
        }
    }
}

// Many tiles in one launch: tile k (tiles contiguous, [k][y][x]) is max-composited at
// x0 + k*step.  Only tiles with k % stride_k == phase are touched, so that overlapping
// neighbours (step < width) are handled by separate launches without a race.
__global__ void fsea_stitch_tiles_kernel(uint8_t *dst, const uint8_t *tiles, uint32_t n_tiles, uint32_t x0,
                                         uint32_t step, uint32_t width, uint32_t height, uint32_t dst_stride,
                                         uint32_t stride_k, uint32_t phase) {
    const uint32_t k = (blockIdx.z * stride_k) + phase;
    if (k >= n_tiles) return;
    const uint32_t x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (x4 >= width) return;
    for (uint32_t y = blockIdx.y; y < height; y += gridDim.y) {
        uint8_t *d = dst + (size_t)y * dst_stride + x0 + (size_t)k * step + x4;
        const uint8_t *s = tiles + ((size_t)k * height + y) * width + x4;
        const bool vec = (x4 + 4 <= width) && ((reinterpret_cast<uintptr_t>(d) & 3) == 0) &&
                         ((reinterpret_cast<uintptr_t>(s) & 3) == 0);
        if (vec) {
            const uint32_t a = *reinterpret_cast<const uint32_t *>(d);
            const uint32_t b = *reinterpret_cast<const uint32_t *>(s);
            uint32_t r = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t ab = (a >> (8 * q)) & 0xff, bb = (b >> (8 * q)) & 0xff;
                r |= (ab > bb ? ab : bb) << (8 * q);
            }
            *reinterpret_cast<uint32_t *>(d) = r;
        } else {
            for (uint32_t q = 0; q < 4 && x4 + q < width; ++q) d[q] = d[q] > s[q] ? d[q] : s[q];
        }
    }
}

__global__ void fsea_sum_f32_kernel(const float *x, size_t n, double *acc) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        s += (double)x[i];
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ double part[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) part[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tot += part[i];
        atomicAdd(acc, tot);
    }
}

// nrf_fft_shift on a device-resident history (src/nrf.c:569-596): every row moved by `shift` bins,
// vacated bins zero; out of place (src and dst are the two halves of the ring's ping-pong storage).
__global__ void fsea_history_shift_kernel(const float *src, float *dst, int n, size_t total, int shift) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % (size_t)n);
        const int from = x + shift;
        dst[i] = (from >= 0 && from < n) ? src[i - (size_t)x + (size_t)from] : 0.0f;
    }
}

__global__ void fsea_f64_to_f32_kernel(const double *in, float *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        out[i] = (float)in[i];
    }
}


}  // namespace

namespace {

// the kernel raw int8 input (flip) launches on a plan of a size with kernels of its own
std::string pow2_kernel_name(const fsea_plan *p) {
    const fsea::KernelEntry *e = p->entry;
    const bool windowed = p->window_form != 0;
    int k = pick_kind(fsea::IN_U8, p->mode, 1);
    if (windowed) {
        k = windowed_kind(k);
        if (!e->fn[k]) k = fsea::K_U8_WIN;
    }
    if (!e->fn[k]) k = fsea::K_U8;
    const int half = windowed ? fsea::K_U8_MAG_HALF_WIN : fsea::K_U8_MAG_HALF;
    if ((k == fsea::K_U8_MAG || k == fsea::K_U8_MAG_WIN) && e->fn[half] && 2 * (size_t)p->hop == (size_t)p->n && !p->no_half_overlap) {
        k = half;  // 50 %-overlapped frames run the half-overlap kernel
    }
    return e->name[k] ? e->name[k] : "";
}

int ensure(void **ptr, size_t *cap, size_t need) {
    if (*cap >= need) return FSEA_OK;
    if (*ptr) {
        FSEA_HIP(hipFree(*ptr));
        *ptr = nullptr;
        *cap = 0;
    }
    size_t want = need + need / 4 + 4096;
    FSEA_HIP(hipMalloc(ptr, want));
    *cap = want;
    return FSEA_OK;
}

constexpr size_t FSEA_ZERO_COPY_MAX = 256 * 1024;  // in + out bytes up to which the staging is mapped host memory

int ensure_pinned(void **ptr, size_t *cap, size_t need) {
    if (*cap >= need) return FSEA_OK;
    if (*ptr) {
        FSEA_HIP(hipHostFree(*ptr));
        *ptr = nullptr;
        *cap = 0;
    }
    const size_t want = need < 65536 ? 65536 : need;
    FSEA_HIP(hipHostMalloc(ptr, want, hipHostMallocMapped));
    *cap = want;
    return FSEA_OK;
}

unsigned grid_for(const fsea_plan *p, const fsea::KernelEntry *e, int occ, size_t n_frames) {
    const size_t units = (n_frames + e->fpw - 1) / e->fpw;
    size_t g = (size_t)p->num_cu * (size_t)(occ > 0 ? occ : 1);
    if (units < g) g = units;
    if (g >= 8) g &= ~(size_t)7;  // the kernel's XCD-aware frame mapping wants a multiple of 8
    if (g == 0) g = 1;
    return (unsigned)g;
}

// The ticket-counter slot for a launch on stream `s` that uses the counters (see fsea_plan::d_ctr).
//  * A stream keeps its slot: launches on one stream run in order and the last workgroup of a launch zeroes the slot.
//  * A slot is handed to another stream once the event recorded behind its last launch has completed (least recently
//    used first), so a plan may see any number of short-lived streams over its lifetime; with FSEA_CTR_SLOTS launches in
//    flight on as many streams, the call waits for the oldest of them instead of failing.
//  * A stream handle that matches a slot whose last launch is still running is either that stream (already ordered) or a
//    new stream that got a destroyed stream's handle: the launch is ordered behind the slot's event either way.
//  * hipStreamPerThread names a different stream in every thread: it never keeps a slot, every launch takes a free one.
//  * While `s` is being captured into a graph nothing but the kernel may be enqueued: no event; the slot then stays
//    with that stream handle for good (replay one instance of such a graph at a time, include/fsea.h).
// Returns null only if an event cannot be created or every slot belongs to a captured stream.
unsigned *counter_slot(fsea_plan *p, hipStream_t s, int *index, bool *record) {
    std::lock_guard<std::mutex> lock(p->slot_mu);
    bool capturing = false;
    if (s != nullptr) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) == hipSuccess) capturing = (st == hipStreamCaptureStatusActive);
        else (void)hipGetLastError();
    }
    const bool anonymous = (s == hipStreamPerThread);
    int pick = -1;
    if (!anonymous) {
        for (unsigned i = 0; i < FSEA_CTR_SLOTS; ++i) {
            if (p->slots[i].used && !p->slots[i].anonymous && p->slots[i].stream == s) pick = (int)i;
        }
        if (pick >= 0 && p->slots[pick].pending && !capturing) {
            if (hipStreamWaitEvent(s, p->slots[pick].ev, 0) != hipSuccess) (void)hipGetLastError();
        }
    }
    if (pick < 0) {
        for (unsigned i = 0; i < FSEA_CTR_SLOTS && pick < 0; ++i) {
            if (!p->slots[i].used) pick = (int)i;
        }
        int oldest = -1;
        for (unsigned i = 0; i < FSEA_CTR_SLOTS && pick < 0; ++i) {
            fsea_plan::CtrSlot &c = p->slots[i];
            if (c.captured || c.launching) continue;  // launching: claimed by another host thread whose kernel is being enqueued
            if (!c.pending || hipEventQuery(c.ev) == hipSuccess) {
                if (pick < 0 || c.seq < p->slots[pick].seq) pick = (int)i;
            } else {
                (void)hipGetLastError();  // hipErrorNotReady is not an error
                if (oldest < 0 || c.seq < p->slots[oldest].seq) oldest = (int)i;
            }
        }
        if (pick < 0 && oldest >= 0) {  // FSEA_CTR_SLOTS launches in flight: wait for the oldest one
            if (hipEventSynchronize(p->slots[oldest].ev) != hipSuccess) return nullptr;
            pick = oldest;
        }
        if (pick < 0) return nullptr;
        p->slots[pick].used = true;
        p->slots[pick].pending = false;
        p->slots[pick].anonymous = anonymous;
        p->slots[pick].stream = s;
    }
    fsea_plan::CtrSlot &c = p->slots[pick];
    // A slot a capture has used stays reserved for its stream until fsea_plan_reset: an instantiated hipGraph has the slot's
    // counter address baked into its kernel node, so recycling it to another stream (round 4 did, once the capturing stream
    // launched un-captured again) would let a replay share one counter with that stream's launches.  The
    // capturing stream itself keeps using it for ordinary launches, in stream order with a replay on that stream.
    if (capturing) c.captured = true;
    if (!capturing && !c.ev) {
        if (hipEventCreateWithFlags(&c.ev, hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    c.seq = ++p->slot_seq;
    c.launching = true;  // until launch_pow2 has enqueued the kernel and recorded the slot's event
    *index = pick;
    *record = !capturing;
    return p->d_ctr + (size_t)FSEA_CTR_WORDS * (size_t)pick;
}

int launch_pow2(fsea_plan *p, int in_kind, const void *d_in, size_t n_frames, int flip, int mode, void *d_out,
                hipStream_t s, double rot_delta, double rot_phase0, const TileLayout *tiles);

}  // namespace

namespace fsea_detail {
int launch(fsea_plan *p, int in_kind, const void *d_in, size_t n_frames, int flip, int mode, void *d_out,
           hipStream_t s, double rot_delta, double rot_phase0, const TileLayout *tiles) {
    if (p->blu_m || p->fs_n1) {
        if (in_kind == fsea::IN_U8_ROT || tiles) {
            return fail(FSEA_EINVAL, "fft_size %d runs through %s: the frequency-shifted and the tiled "
                                     "entry points exist for the power-of-two sizes from 32 to 16384 only", p->n,
                        p->blu_m ? "Bluestein's algorithm" : "the four-step decomposition");
        }
        if (s != nullptr) {  // five to ten launches through the plan's own work buffers, ordered by events: not capturable
            hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive) {
                return fail(FSEA_EINVAL, "fft_size %d runs through %s, whose launches cannot be captured into a graph (include/fsea.h); "
                                         "only the powers of two from 32 to 16384 can", p->n,
                            p->blu_m ? "Bluestein's algorithm" : "the four-step decomposition");
            }
            (void)hipGetLastError();
        }
        std::lock_guard<std::mutex> lock(p->work_mu);
        if (!p->work_ev) FSEA_HIP(hipEventCreateWithFlags(&p->work_ev, hipEventDisableTiming));
        if (p->work_pending) FSEA_HIP(hipStreamWaitEvent(s, p->work_ev, 0));
        const int rc = p->blu_m ? blu_launch(p, in_kind, d_in, n_frames, flip, mode, d_out, s)
                                : fs_launch(p, in_kind, d_in, n_frames, flip, mode, d_out, s);
        FSEA_HIP(hipEventRecord(p->work_ev, s));
        p->work_pending = true;
        return rc;
    }
    return launch_pow2(p, in_kind, d_in, n_frames, flip, mode, d_out, s, rot_delta, rot_phase0, tiles);
}
}  // namespace fsea_detail

namespace {

int launch_pow2(fsea_plan *p, int in_kind, const void *d_in, size_t n_frames, int flip, int mode, void *d_out,
                hipStream_t s, double rot_delta, double rot_phase0, const TileLayout *tiles) {
    if (n_frames == 0) return FSEA_OK;
    int kind = pick_kind(in_kind, mode, flip);
    const fsea::KernelEntry *e = p->entry;
    const bool windowed = p->window_form != 0;
    if (windowed) {
        if (in_kind == fsea::IN_U8) {
            kind = windowed_kind(kind);
            if (!e->fn[kind]) kind = fsea::K_U8_WIN;
        } else {
            // the nrf_freq_shifter -> nrf_fft chain and the F64 branch of nrf_fft_process (src/nrf.c:607-612) with the taper
            kind = in_kind == fsea::IN_F32 ? fsea::K_F32_WIN : fsea::K_U8_ROT_WIN;
        }
    }
    // tuning variants carry the u8 MAG and run-time-mode kernels only: their pixel modes run the latter
    if (!e->fn[kind] && (kind == fsea::K_U8_DB5 || kind == fsea::K_U8_DB10)) kind = fsea::K_U8;
    if (!e->fn[kind]) {
        return fail(FSEA_EINVAL, "kernel variant '%s' has no entry point for this input kind", e->variant);
    }
    fsea::FftArgs a;
    a.win = p->d_win;
    a.win_dc = p->d_win_dc;
    a.win_offset = p->window_form == 2 ? 1u : 0u;
    a.rot_delta = rot_delta;
    a.rot_phase0 = rot_phase0;
    if (in_kind == fsea::IN_U8_ROT) {
        fsea::TwPair rows[32];
        fsea::build_rotation_rows(p->n, e->radix[0], rot_delta, rows);
        for (int r = 0; r < 32; ++r) a.rot_row[r] = fsea::cf{rows[r].re, rows[r].im};
    }
    a.in = d_in;
    a.out = d_out;
    a.n_frames = n_frames;
    a.hop = (size_t)p->hop;
    a.xormask = flip ? 0u : 0x80808080u;
    a.mode = mode;
    a.trace = p->d_trace;
    for (int i = 0; i < 4; ++i) a.tw[i] = p->d_tw + p->tw_off[i];
    a.tw_small = p->d_tw;
    a.tw_def = p->d_tw + p->tw_def_off;
    // units per workgroup of this launch: few -> static interleave, many -> ticket pools (FftArgs::dynamic_units)
    {
        const unsigned grid = grid_for(p, e, p->occ[kind], n_frames);
        const size_t n_units = (n_frames + (size_t)e->fpw - 1) / (size_t)e->fpw;
        a.dynamic_units = p->units_policy == FSEA_UNITS_TICKETS  ? 1u
                          : p->units_policy == FSEA_UNITS_STATIC ? 0u
                                                                 : (n_units > (size_t)FSEA_STATIC_UNITS_PER_WG * grid ? 1u : 0u);
    }
    if (tiles) {
        a.tile_rows = tiles->rows;
        a.pitch_row = tiles->pitch_row;
        a.pitch_tile = tiles->pitch_tile;
        a.out_span = tiles->span;
    }
    // 50 %-overlapped frames of the nrf_fft_process kind (raw int8, MAG rows) at the sizes with one frame per workgroup:
    // the half-overlap kernel, runs of consecutive frames per workgroup (every sample loaded once), static units
    const int half_kind = windowed ? fsea::K_U8_MAG_HALF_WIN : fsea::K_U8_MAG_HALF;
    if ((kind == fsea::K_U8_MAG || kind == fsea::K_U8_MAG_WIN) && e->fn[half_kind] && !tiles &&
        2 * (size_t)p->hop == (size_t)p->n && n_frames >= 2 && !p->no_half_overlap) {
        const size_t wgs = (size_t)p->num_cu * (size_t)(p->occ[half_kind] > 0 ? p->occ[half_kind] : 1);
        size_t run = (n_frames + wgs - 1) / wgs;  // frames per run: long enough to reuse most halves, short enough that every workgroup gets some
        if (run > (size_t)p->half_run_max) run = (size_t)p->half_run_max;
        if (run < 1) run = 1;
        a.run_len = (uint32_t)run;
        a.dynamic_units = 0;
        a.ctr = p->d_ctr;
        const size_t units = (n_frames + run - 1) / run;
        size_t g = wgs < units ? wgs : units;
        if (g >= 8) g &= ~(size_t)7;
        if (windowed) e->launch_win(half_kind, a, (unsigned)g, s);
        else e->launch_half(a, (unsigned)g, s);
        FSEA_HIP(hipGetLastError());
        return FSEA_OK;
    }
    // only a launch that hands its frames out by the ticket pools needs a counter slot (single-wave sizes and short
    // launches never touch the counters)
    int slot = -1;
    bool record = false;
    a.ctr = p->d_ctr;
    if (e->counters == 2 || (e->counters == 1 && a.dynamic_units != 0)) {
        a.ctr = counter_slot(p, s, &slot, &record);
        if (!a.ctr) {
            return fail(FSEA_EHIP, "no ticket-counter slot for this launch: all %u slots of the plan are reserved by streams with captured "
                                   "launches (fsea_plan_release_stream once a stream's graphs are destroyed, or fsea_plan_reset), or an "
                                   "event could not be created", FSEA_CTR_SLOTS);
        }
    }
    if (kind >= fsea::K_U8_MAG_WIN) e->launch_win(kind, a, grid_for(p, e, p->occ[kind], n_frames), s);
    else e->launch(kind, a, grid_for(p, e, p->occ[kind], n_frames), s);
    const hipError_t launched = hipGetLastError();
    if (slot >= 0) {
        std::lock_guard<std::mutex> lock(p->slot_mu);
        p->slots[slot].launching = false;
        if (record && launched == hipSuccess) {
            FSEA_HIP(hipEventRecord(p->slots[slot].ev, s));
            p->slots[slot].pending = true;
        }
    }
    if (launched != hipSuccess) return fail(FSEA_EHIP, "kernel launch failed: %s", hipGetErrorString(launched));
    return FSEA_OK;
}

int check_exec_args(const fsea_plan *plan, const void *in, const void *out, size_t align) {
    if (!plan) return fail(FSEA_EINVAL, "plan is NULL");
    if (!in || !out) return fail(FSEA_EINVAL, "NULL buffer");
    if (reinterpret_cast<uintptr_t>(in) % align) return fail(FSEA_EINVAL, "input pointer must be %zu-byte aligned", align);
    if (reinterpret_cast<uintptr_t>(out) % 16) return fail(FSEA_EINVAL, "output pointer must be 16-byte aligned");
    return FSEA_OK;
}

}  // namespace

extern "C" {

const char *fsea_last_error_string(void) { return g_last_error.c_str(); }

int fsea_device_count(int *count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (count) *count = (e == hipSuccess) ? n : 0;
    if (e != hipSuccess || n <= 0) return fail(FSEA_ENODEVICE, "no HIP device: %s", hipGetErrorString(e));
    return FSEA_OK;
}

static int create_plan(fsea_plan **out, int fft_size, int hop, int mode, int device, const char *variant) {
    if (!out) return fail(FSEA_EINVAL, "plan out-pointer is NULL");
    *out = nullptr;
    if (mode < FSEA_MODE_MAG_F32 || mode > FSEA_MODE_DB_F32) return fail(FSEA_EINVAL, "unknown mode %d", mode);
    const fsea::KernelEntry *e = find_entry(fft_size, variant);
    int blu_m = 0, fs_n1 = 0, fs_n2 = 0;
    if (!e) {
        if (variant && variant[0]) {
            return fail(FSEA_EINVAL, "no kernel variant '%s' for fft_size %d", variant, fft_size);
        }
        // a size FFTW takes and no kernel has: a power of two above 16384 in two passes of the kernels (four-step), anything
        // else by Bluestein's algorithm on a power of two m >= 2n - 1
        if (fourstep_split(fft_size, &fs_n1, &fs_n2)) {
            e = find_entry(fs_n1, "");
        } else {
            blu_m = bluestein_m(fft_size);
            if (blu_m) e = find_entry(blu_m <= 16384 ? blu_m : 16384, "");
        }
        if (!e) {
            return fail(FSEA_EINVAL, "unsupported fft_size %d: the gfx950 kernels cover powers of two in [32, 16384] directly, larger "
                                     "powers of two up to %d in two passes, and every other size from 2 to %d through Bluestein's "
                                     "algorithm", fft_size, FSEA_MAX_FFT_SIZE, (FSEA_MAX_FFT_SIZE + 1) / 2);
        }
    }
    if (hop <= 0 || (!blu_m && !fs_n1 && (hop % 8) != 0)) {
        return fail(FSEA_EINVAL, "hop must be a positive multiple of 8 (any positive hop for the sizes without a kernel of their "
                                 "own); got %d", hop);
    }
    int count = 0;
    hipError_t ce = hipGetDeviceCount(&count);
    if (ce != hipSuccess || count <= 0) {
        return fail(FSEA_ENODEVICE, "no HIP device available (%s); libfsea_hip has no CPU fallback",
                    hipGetErrorString(ce));
    }
    if (device < 0 || device >= count) return fail(FSEA_EINVAL, "device %d out of range [0,%d)", device, count);
    FSEA_ON_DEVICE(device);
    hipDeviceProp_t prop;
    FSEA_HIP(hipGetDeviceProperties(&prop, device));

    fsea_plan *p = new (std::nothrow) fsea_plan();
    if (!p) return fail(FSEA_ENOMEM, "out of host memory");
    p->n = fft_size;
    p->hop = hop;
    p->mode = mode;
    p->device = device;
    p->entry = e;
    p->blu_m = blu_m;
    p->fs_n1 = fs_n1;
    p->fs_n2 = fs_n2;
    p->num_cu = prop.multiProcessorCount;
    p->no_half_overlap = std::getenv("FSEA_NO_HALF_OVERLAP") != nullptr;
    if (const char *hr = std::getenv("FSEA_HALF_RUN_MAX")) {
        const int v = std::atoi(hr);
        if (v >= 1 && v <= 4096) p->half_run_max = v;
    }
    p->kernel_name = pow2_kernel_name(p);
#ifdef FSEA_TUNE
    if (std::getenv("FSEA_TRACE")) {
        if (hipMalloc(reinterpret_cast<void **>(&p->d_trace), 4096 * 32 * sizeof(unsigned long long)) != hipSuccess) {
            p->d_trace = nullptr;
        }
    }
#endif

    std::vector<fsea::TwPair> tw;
    fsea::build_twiddles(e->np, e->radix, tw, p->tw_off);
    {
        std::vector<fsea::TwPair> def;
        fsea::build_deferred_table(e->radix[0], e->radix[1], def);
        if (tw.size() & 1) tw.push_back(fsea::TwPair{0.f, 0.f});
        p->tw_def_off = tw.size();
        tw.insert(tw.end(), def.begin(), def.end());
    }
    static_assert(sizeof(fsea::TwPair) == sizeof(fsea::cf), "twiddle layout");
    hipError_t he = hipMalloc(reinterpret_cast<void **>(&p->d_tw), tw.size() * sizeof(fsea::cf));
    if (he == hipSuccess) he = hipMemcpy(p->d_tw, tw.data(), tw.size() * sizeof(fsea::cf), hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void **>(&p->d_acc), sizeof(double));
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void **>(&p->d_ctr), FSEA_CTR_SLOTS * FSEA_CTR_WORDS * sizeof(unsigned));
    if (he == hipSuccess) he = hipMemset(p->d_ctr, 0, FSEA_CTR_SLOTS * FSEA_CTR_WORDS * sizeof(unsigned));
    // the memset runs on the null stream; the caller's (possibly non-blocking) streams are not ordered
    // behind it, so it is complete before the plan is handed out
    if (he == hipSuccess) he = hipDeviceSynchronize();
    if (he == hipSuccess) he = hipEventCreate(&p->ev0);
    if (he == hipSuccess) he = hipEventCreate(&p->ev1);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&p->s_h2d, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&p->s_d2h, hipStreamNonBlocking);
    for (unsigned c = 0; c < FSEA_HOST_CHUNKS_MAX && he == hipSuccess; ++c) {
        he = hipEventCreateWithFlags(&p->ev_in[c], hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&p->ev_done[c], hipEventDisableTiming);
    }
    for (int k = 0; k < fsea::K_COUNT && he == hipSuccess; ++k) {
        if (e->fn[k]) he = hipOccupancyMaxActiveBlocksPerMultiprocessor(&p->occ[k], e->fn[k], e->wg, 0);
    }
    if (he != hipSuccess) {
        int rc = fail(FSEA_EHIP, "plan setup failed: %s", hipGetErrorString(he));
        fsea_plan_destroy(p);
        return rc;
    }
    if (blu_m) {
        int rc = create_plan(&p->blu_inner, blu_m, blu_m, FSEA_MODE_COMPLEX_F32, device, "");
        if (rc == FSEA_OK) rc = blu_setup(p);
        if (rc != FSEA_OK) {
            fsea_plan_destroy(p);
            return rc;
        }
        p->kernel_name = std::string("bluestein(") +
                         (p->blu_inner->fs_n1 ? p->blu_inner->kernel_name : std::string(p->blu_inner->entry->name[fsea::K_F32])) + " x2)";
    }
    if (fs_n1) {
        int rc = create_plan(&p->fs_inner1, fs_n1, fs_n1, FSEA_MODE_COMPLEX_F32, device, "");
        if (rc == FSEA_OK) rc = create_plan(&p->fs_inner2, fs_n2, fs_n2, FSEA_MODE_COMPLEX_F32, device, "");
        if (rc == FSEA_OK) rc = fs_setup(p);
        if (rc != FSEA_OK) {
            fsea_plan_destroy(p);
            return rc;
        }
        p->kernel_name = std::string("fourstep(") + p->fs_inner1->entry->name[fsea::K_F32] + ", " + p->fs_inner2->entry->name[fsea::K_F32] + ")";
    }
    *out = p;
    return FSEA_OK;
}

// The configuration a plan of (fft_size, mode) takes where the modes of a size prefer different radix orders
// (fsea_configs.h: FSEA_CFG_256_ROWS, FSEA_CFG_512_PX, FSEA_CFG_1024_RT); "" = the size's first configuration.
static const char *preferred_variant(int fft_size, int mode) {
    const bool f32_rows = mode == FSEA_MODE_MAG_F32 || mode == FSEA_MODE_MAG_NODC_F32 || mode == FSEA_MODE_DB_F32;
    const bool pixels = mode == FSEA_MODE_DB10_U8 || mode == FSEA_MODE_DB5_U8_DCFIX;
    if (fft_size == 256 && f32_rows) return "rows";
    if (fft_size == 512 && pixels) return "px";
    // 1024 points: the modes only the run-time-mode kernel serves (COMPLEX_F32, MAG_NODC_F32, DB_F32)
    if (fft_size == 1024 && (mode == FSEA_MODE_COMPLEX_F32 || mode == FSEA_MODE_MAG_NODC_F32 || mode == FSEA_MODE_DB_F32)) return "rt";
    return "";
}

int fsea_plan_create(fsea_plan **out, int fft_size, int hop, int mode, int device) {
    const char *variant = std::getenv("FSEA_ONE_CONFIG_PER_SIZE") ? "" : preferred_variant(fft_size, mode);  // (A/B measurements)
    if (variant[0] && !find_entry(fft_size, variant)) variant = "";
    return create_plan(out, fft_size, hop, mode, device, variant);
}

// Recovery: waits for the device and zeroes every ticket-counter slot (a launch that was aborted
// leaves its slot non-zero, which would make later launches on that stream skip or repeat frames).
int fsea_plan_reset(fsea_plan *p) {
    if (!p) return fail(FSEA_EINVAL, "plan is NULL");
    FSEA_ON_DEVICE(p->device);
    FSEA_HIP(hipDeviceSynchronize());
    FSEA_HIP(hipMemset(p->d_ctr, 0, FSEA_CTR_SLOTS * FSEA_CTR_WORDS * sizeof(unsigned)));
    FSEA_HIP(hipDeviceSynchronize());
    {
        std::lock_guard<std::mutex> lock(p->slot_mu);
        for (auto &c : p->slots) c.used = c.pending = c.anonymous = c.captured = c.launching = false;
    }
    return FSEA_OK;
}

// Gives back the ticket-counter slot `stream` holds in this plan -- in particular one reserved by a captured launch -- once
// the graphs captured on that stream are destroyed (an application that captures on short-lived streams would
// otherwise run out of the 64 slots).  Waits for the slot's last un-captured launch; the slot's counters are left as every
// finished launch leaves them (zero).  FSEA_OK also when the stream holds no slot.
int fsea_plan_release_stream(fsea_plan *p, void *stream) {
    if (!p) return fail(FSEA_EINVAL, "plan is NULL");
    FSEA_ON_DEVICE(p->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (s != nullptr) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive) {
            return fail(FSEA_EINVAL, "fsea_plan_release_stream: the stream is being captured");
        }
        (void)hipGetLastError();
    }
    std::lock_guard<std::mutex> lock(p->slot_mu);
    for (auto &c : p->slots) {
        if (!c.used || c.anonymous || c.stream != s) continue;
        if (c.launching) return fail(FSEA_EINVAL, "fsea_plan_release_stream: another thread is launching on this stream");
        if (c.pending && c.ev) FSEA_HIP(hipEventSynchronize(c.ev));
        c.used = c.pending = c.captured = false;
    }
    return FSEA_OK;
}

int fsea_plan_destroy(fsea_plan *p) {
    if (!p) return FSEA_OK;
    DeviceGuard device_guard_(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->blu_inner) (void)fsea_plan_destroy(p->blu_inner);
    if (p->work_ev) (void)hipEventDestroy(p->work_ev);
    if (p->fs_inner1) (void)fsea_plan_destroy(p->fs_inner1);
    if (p->fs_inner2) (void)fsea_plan_destroy(p->fs_inner2);
    if (p->d_fs_tw) (void)hipFree(p->d_fs_tw);
    if (p->d_blu_chirp) (void)hipFree(p->d_blu_chirp);
    if (p->d_blu_dc) (void)hipFree(p->d_blu_dc);
    if (p->d_blu_bfft) (void)hipFree(p->d_blu_bfft);
    if (p->d_blu_work[0]) (void)hipFree(p->d_blu_work[0]);
    if (p->d_blu_work[1]) (void)hipFree(p->d_blu_work[1]);
    if (p->d_tw) (void)hipFree(p->d_tw);
    if (p->d_win) (void)hipFree(p->d_win);
    if (p->d_win_dc) (void)hipFree(p->d_win_dc);
    if (p->d_in) (void)hipFree(p->d_in);
    if (p->d_out) (void)hipFree(p->d_out);
    if (p->d_aux) (void)hipFree(p->d_aux);
    if (p->d_acc) (void)hipFree(p->d_acc);
    if (p->h_in) (void)hipHostFree(p->h_in);
    if (p->h_out) (void)hipHostFree(p->h_out);
    if (p->d_trace) (void)hipFree(p->d_trace);
    if (p->d_ctr) (void)hipFree(p->d_ctr);
    for (auto &c : p->slots) {
        if (c.ev) (void)hipEventDestroy(c.ev);
    }
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    for (unsigned c = 0; c < FSEA_HOST_CHUNKS_MAX; ++c) {
        if (p->ev_in[c]) (void)hipEventDestroy(p->ev_in[c]);
        if (p->ev_done[c]) (void)hipEventDestroy(p->ev_done[c]);
    }
    if (p->s_h2d) (void)hipStreamDestroy(p->s_h2d);
    if (p->s_d2h) (void)hipStreamDestroy(p->s_d2h);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return FSEA_OK;
}

size_t fsea_plan_row_bytes(const fsea_plan *p) { return p ? (size_t)p->n * mode_elem_bytes(p->mode) : 0; }
int fsea_plan_fft_size(const fsea_plan *p) { return p ? p->n : 0; }
const char *fsea_plan_kernel_name(const fsea_plan *p) {
    if (!p) return "";
    return (p->window_form != 0 && !p->kernel_name_win.empty()) ? p->kernel_name_win.c_str() : p->kernel_name.c_str();
}

int fsea_plan_set_unit_distribution(fsea_plan *p, int policy) {
    if (!p) return fail(FSEA_EINVAL, "plan is NULL");
    if (policy != FSEA_UNITS_AUTO && policy != FSEA_UNITS_STATIC && policy != FSEA_UNITS_TICKETS) {
        return fail(FSEA_EINVAL, "unknown unit distribution %d", policy);
    }
    p->units_policy = policy;
    return FSEA_OK;
}

int fsea_window_fill(int kind, int n, float *w) {
    static const double coef[6][5] = {{1.0, 0, 0, 0, 0},
                                      {0.5, 0.5, 0, 0, 0},
                                      {0.54, 0.46, 0, 0, 0},
                                      {0.42, 0.5, 0.08, 0, 0},
                                      {0.35875, 0.48829, 0.14128, 0.01168, 0},
                                      {0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368}};
    if (kind < FSEA_WINDOW_RECT || kind > FSEA_WINDOW_FLATTOP) return fail(FSEA_EINVAL, "unknown window kind %d", kind);
    if (n < 1 || !w) return fail(FSEA_EINVAL, "fsea_window_fill: n >= 1 and a buffer of n floats");
    const double tau = 6.283185307179586476925286766559;
    for (int j = 0; j < n; ++j) {
        double v = 0.0, sign = 1.0;
        for (int k = 0; k < 5; ++k) {
            if (coef[kind][k] != 0.0) v += sign * coef[kind][k] * std::cos(tau * (double)k * (double)j / (double)n);
            sign = -sign;
        }
        w[j] = (float)v;
    }
    return FSEA_OK;
}

int fsea_plan_window_form(const fsea_plan *p) { return p ? p->window_form : 0; }

int fsea_plan_set_window(fsea_plan *p, const float *w) {
    if (!p) return fail(FSEA_EINVAL, "plan is NULL");
    if (p->blu_m || p->fs_n1) {
        return fail(FSEA_EINVAL, "fft_size %d has no kernel of its own (Bluestein / four-step path): the taper window is fused into "
                                 "the kernels of the powers of two from 32 to 16384", p->n);
    }
    const fsea::KernelEntry *e = p->entry;
    FSEA_ON_DEVICE(p->device);
    FSEA_HIP(hipDeviceSynchronize());  // no launch of this plan may still be reading the tables that are replaced
    if (!w) {
        p->window_form = 0;
        return FSEA_OK;
    }
    if (!e->fn[fsea::K_U8_WIN] || !e->fn[fsea::K_U8_MAG_WIN]) {
        return fail(FSEA_EINVAL, "kernel variant '%s' of size %d has no windowed kernels", e->variant, p->n);
    }
    const int n = p->n;
    for (int j = 0; j < n; ++j) {
        if (!std::isfinite(w[j])) return fail(FSEA_EINVAL, "window weight %d is not finite", j);
    }
    std::vector<float> perm;
    std::vector<fsea::TwPair> dc;
    const int form = fsea::build_window_tables(n, e->t, e->radix[0], e->radix[e->np - 1], w, perm, dc);
    if (!p->d_win) FSEA_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_win), (size_t)n * sizeof(float)));
    if (!p->d_win_dc) FSEA_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_win_dc), dc.size() * sizeof(fsea::cf)));
    FSEA_HIP(hipMemcpy(p->d_win, perm.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    FSEA_HIP(hipMemcpy(p->d_win_dc, dc.data(), dc.size() * sizeof(fsea::cf), hipMemcpyHostToDevice));
    p->window_form = form;
    // computed once: everything pow2_kernel_name reads (entry, mode, hop, no_half_overlap) is fixed at plan creation
    // (FSEA_NO_HALF_OVERLAP is read there, not per launch), so the windowed name cannot go stale
    if (p->kernel_name_win.empty()) p->kernel_name_win = pow2_kernel_name(p);
    return FSEA_OK;
}

int fsea_plan_grid(const fsea_plan *p, size_t n_frames, unsigned *grid, unsigned *block, size_t *lds_bytes) {
    if (!p) return fail(FSEA_EINVAL, "plan is NULL");
    int k = pick_kind(fsea::IN_U8, p->mode, 1);
    if (p->window_form) k = windowed_kind(k);
    if (!p->entry->fn[k]) k = p->window_form ? fsea::K_U8_WIN : fsea::K_U8;
    if (grid) *grid = grid_for(p, p->entry, p->occ[k], n_frames);
    if (block) *block = (unsigned)p->entry->wg;
    // a plan with a taper launches the *_WIN kernels, whose static LDS carries the DC table on top
    if (lds_bytes) {
        *lds_bytes = p->entry->lds_bytes;
        if (p->window_form != 0 && p->entry->lds_bytes_win) {
            *lds_bytes = (k == fsea::K_U8_MAG_WIN || k == fsea::K_U8_MAG_HALF_WIN) ? p->entry->lds_bytes_win : p->entry->lds_bytes_win_other;
        }
    }
    return FSEA_OK;
}

int fsea_exec_u8_device(fsea_plan *p, const void *d_iq, size_t n_frames, int flip, void *d_out, void *stream) {
    int rc = check_exec_args(p, d_iq, d_out, 16);
    if (rc) return rc;
    FSEA_ON_DEVICE(p->device);
    return launch(p, fsea::IN_U8, d_iq, n_frames, flip, p->mode, d_out,
                  static_cast<hipStream_t>(stream));
}

int fsea_exec_u8_tiled_device(fsea_plan *p, const void *d_iq, size_t n_frames, int flip, void *d_image,
                              size_t image_rows, size_t image_stride, size_t first_x, size_t tile_rows, size_t tile_step,
                              void *stream) {
    int rc = check_exec_args(p, d_iq, d_image, 16);
    if (rc) return rc;
    if (n_frames == 0) return FSEA_OK;
    const size_t n = (size_t)p->n, esz = fsea_plan_row_bytes(p) / n;
    const int fpw = p->entry->fpw;
    if (tile_rows == 0 || tile_rows > image_rows) return fail(FSEA_EINVAL, "tile_rows must be in [1, image_rows]");
    if (tile_rows % (size_t)fpw != 0) {
        return fail(FSEA_EINVAL, "tile_rows must be a multiple of %d at fft_size %d (frames per workgroup)", fpw, p->n);
    }
    if (n_frames % tile_rows != 0) return fail(FSEA_EINVAL, "n_frames must be whole tiles (a multiple of tile_rows)");
    if (tile_step < n) return fail(FSEA_EINVAL, "tile_step must be at least fft_size: tiles are written, not max-composited");
    if ((image_stride | first_x | tile_step) % 4 != 0) {
        return fail(FSEA_EINVAL, "image_stride, first_x and tile_step must be multiples of 4 elements");
    }
    const size_t n_tiles = n_frames / tile_rows;
    if (first_x + (n_tiles - 1) * tile_step + n > image_stride) return fail(FSEA_EINVAL, "tiles leave the image row");
    const size_t span = image_rows * image_stride - first_x;
    if (n_frames > 0xffffffffull || image_stride > 0xffffffffull || tile_step > 0xffffffffull ||
        ((size_t)(fpw - 1) * image_stride + n) * esz > 0xffffffffull) {
        return fail(FSEA_EINVAL, "image geometry exceeds the kernel's 32-bit row offsets");
    }
    TileLayout t;
    t.rows = (uint32_t)tile_rows;
    t.pitch_row = (uint32_t)image_stride;
    t.pitch_tile = (uint32_t)tile_step;
    t.span = span;
    FSEA_ON_DEVICE(p->device);
    return launch(p, fsea::IN_U8, d_iq, n_frames, flip, p->mode, static_cast<char *>(d_image) + first_x * esz,
                  static_cast<hipStream_t>(stream), 0.0, 0.0, &t);
}

}  // extern "C"

namespace {

// Pins the caller's buffer in place for the duration of one call, if the runtime lets us: copies from / to pinned pages
// are truly asynchronous (the two directions overlap: 2.56 instead of 3.57 ms for 64 MiB in + 128 MiB out on this box,
// profiles/r03_host_path.txt), pageable ones are staged by the runtime and return when done.  Memory that is pinned
// already (hipHostMalloc, fsea_host_alloc, registered by the caller) is left alone.
// Two host threads may hand the same buffer to two plans at once (one capture, two transform sizes): the registration is
// shared and counted, so that the first call to finish does not unpin pages the other one's copies are still using.
struct PinRegistry {
    struct Entry {
        void *ptr;
        size_t bytes;
        int users;
    };
    std::mutex mu;
    std::vector<Entry> live;
};
PinRegistry &pin_registry() {
    static PinRegistry r;
    return r;
}

struct PinnedInPlace {
    std::vector<void *> held;  // registrations this call keeps alive: its own, or those of calls in flight that it overlaps
    PinnedInPlace(const void *p, size_t bytes) {
        PinRegistry &r = pin_registry();
        std::lock_guard<std::mutex> lock(r.mu);
        const char *lo = static_cast<const char *>(p), *hi = lo + bytes;
        for (auto &e : r.live) {
            const char *elo = static_cast<const char *>(e.ptr), *ehi = elo + e.bytes;
            if (lo < ehi && elo < hi) {  // pinned (in part) by a call in flight on another thread: its pages must stay
                ++e.users;               // pinned until this call's copies are done as well
                held.push_back(e.ptr);
            }
        }
        if (!held.empty()) return;  // fully covered: asynchronous copies; partly: the runtime stages what is pageable
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type != hipMemoryTypeUnregistered) return;  // pinned or device
        (void)hipGetLastError();
        if (hipHostRegister(const_cast<void *>(p), bytes, hipHostRegisterDefault) == hipSuccess) {
            held.push_back(const_cast<void *>(p));
            r.live.push_back(PinRegistry::Entry{const_cast<void *>(p), bytes, 1});
        } else {
            (void)hipGetLastError();  // read-only mapping, foreign registration, ...: pageable copies still work
        }
    }
    ~PinnedInPlace() {
        if (held.empty()) return;
        PinRegistry &r = pin_registry();
        std::lock_guard<std::mutex> lock(r.mu);
        for (void *ptr : held) {
            for (size_t i = 0; i < r.live.size(); ++i) {
                if (r.live[i].ptr != ptr) continue;
                if (--r.live[i].users == 0) {
                    (void)hipHostUnregister(ptr);
                    r.live.erase(r.live.begin() + (long)i);
                }
                break;
            }
        }
    }
    PinnedInPlace(const PinnedInPlace &) = delete;
    PinnedInPlace &operator=(const PinnedInPlace &) = delete;
};

// Host-buffer execution, pipelined: the batch is cut into chunks of whole frames; chunk c's bytes travel on the copy-in
// stream while chunk c-1 is transformed on the plan's stream and chunk c-2's rows travel back on the copy-out stream
// (events order the three).  The streaming shape of the reference's tools (c/fft-batch.c:54-102: one transfer in, one
// row out) at the granularity a PCIe link wants.  bytes_per_sample: 2 (u8 IQ) or 16 (f64 IQ, narrowed on the device).
int exec_host_pipelined(fsea_plan *p, int in_kind, const void *in, size_t bytes_per_sample, size_t n_frames, int flip,
                        void *out, double rot_delta, double rot_phase0) {
    const size_t n = (size_t)p->n, hop = (size_t)p->hop;
    const size_t row_bytes = fsea_plan_row_bytes(p);
    const size_t n_samples = (n_frames - 1) * hop + n;
    const size_t in_bytes = n_samples * bytes_per_sample, out_bytes = n_frames * row_bytes;
    const bool f64 = bytes_per_sample == 16;
    int rc = ensure(f64 ? &p->d_aux : &p->d_in, f64 ? &p->d_aux_bytes : &p->d_in_bytes, in_bytes);
    if (rc) return rc;
    if (f64) {
        rc = ensure(&p->d_in, &p->d_in_bytes, n_samples * 2 * sizeof(float));
        if (rc) return rc;
    }
    rc = ensure(&p->d_out, &p->d_out_bytes, out_bytes);
    if (rc) return rc;
    // chunks of about 24 MiB (in + out): long enough for the link's full rate, short enough that the first copy-in and
    // the last copy-out (the two pieces nothing overlaps) are a small part of the call
    size_t chunks = (in_bytes + out_bytes) / ((size_t)24 << 20);
    if (chunks < 1) chunks = 1;
    if (chunks > FSEA_HOST_CHUNKS_MAX) chunks = FSEA_HOST_CHUNKS_MAX;
    size_t per = (n_frames + chunks - 1) / chunks;
    const size_t fpw = (size_t)p->entry->fpw;
    per = (per + fpw - 1) / fpw * fpw;  // whole units, so that a chunk boundary never splits a workgroup's frames
    chunks = (n_frames + per - 1) / per;
    PinnedInPlace pin_in(in, in_bytes), pin_out(out, out_bytes);
    auto run = [&]() -> int {
        const char *src = static_cast<const char *>(in);
        char *d_src = static_cast<char *>(f64 ? p->d_aux : p->d_in);
        size_t copied = 0;  // input bytes already on their way
        for (size_t c = 0; c < chunks; ++c) {
            const size_t f0 = c * per, f1 = (f0 + per < n_frames) ? f0 + per : n_frames;
            const size_t need = ((f1 - 1) * hop + n) * bytes_per_sample;  // everything chunk c reads (with its overlap into the next)
            if (need > copied) {
                FSEA_HIP(hipMemcpyAsync(d_src + copied, src + copied, need - copied, hipMemcpyHostToDevice, p->s_h2d));
                copied = need;
            }
            FSEA_HIP(hipEventRecord(p->ev_in[c], p->s_h2d));
            FSEA_HIP(hipStreamWaitEvent(p->stream, p->ev_in[c], 0));
            const char *d_frames = static_cast<const char *>(p->d_in) + f0 * hop * (f64 ? 2 * sizeof(float) : 2);
            if (f64) {
                const size_t v0 = f0 * hop * 2, v1 = ((f1 - 1) * hop + n) * 2;  // doubles of this chunk
                unsigned blocks = (unsigned)((v1 - v0 + 255) / 256);
                if (blocks > 2048) blocks = 2048;
                hipLaunchKernelGGL(fsea_f64_to_f32_kernel, dim3(blocks), dim3(256), 0, p->stream,
                                   static_cast<const double *>(p->d_aux) + v0, static_cast<float *>(p->d_in) + v0, v1 - v0);
            }
            const int lrc = launch(p, in_kind, d_frames, f1 - f0, flip, p->mode, static_cast<char *>(p->d_out) + f0 * row_bytes,
                                   p->stream, rot_delta, rot_phase0 + rot_delta * (double)(f0 * hop));
            if (lrc) return lrc;
            FSEA_HIP(hipEventRecord(p->ev_done[c], p->stream));
            FSEA_HIP(hipStreamWaitEvent(p->s_d2h, p->ev_done[c], 0));
            FSEA_HIP(hipMemcpyAsync(static_cast<char *>(out) + f0 * row_bytes, static_cast<char *>(p->d_out) + f0 * row_bytes,
                                    (f1 - f0) * row_bytes, hipMemcpyDeviceToHost, p->s_d2h));
        }
        FSEA_HIP(hipStreamSynchronize(p->s_d2h));
        return FSEA_OK;
    };
    rc = run();
    if (rc) {  // nothing of this call may still be using the caller's pages when they are unpinned
        (void)hipStreamSynchronize(p->s_h2d);
        (void)hipStreamSynchronize(p->stream);
        (void)hipStreamSynchronize(p->s_d2h);
    }
    return rc;
}

// Common body of the u8 host entry points.
int exec_u8_host(fsea_plan *p, int in_kind, const uint8_t *iq, size_t n_frames, int flip, void *out, double rot_delta,
                 double rot_phase0) {
    if (!p) return fail(FSEA_EINVAL, "plan is NULL");
    if (n_frames == 0) return FSEA_OK;
    if (!iq || !out) return fail(FSEA_EINVAL, "NULL buffer");
    std::lock_guard<std::mutex> lock(p->mu);
    FSEA_ON_DEVICE(p->device);
    const size_t in_bytes = 2 * ((n_frames - 1) * (size_t)p->hop + (size_t)p->n);
    const size_t out_bytes = n_frames * fsea_plan_row_bytes(p);
    if (in_bytes + out_bytes <= FSEA_ZERO_COPY_MAX) {
        int rc = ensure_pinned(&p->h_in, &p->h_in_bytes, in_bytes);
        if (rc) return rc;
        rc = ensure_pinned(&p->h_out, &p->h_out_bytes, out_bytes);
        if (rc) return rc;
        void *d_in = nullptr, *d_out = nullptr;
        FSEA_HIP(hipHostGetDevicePointer(&d_in, p->h_in, 0));
        FSEA_HIP(hipHostGetDevicePointer(&d_out, p->h_out, 0));
        std::memcpy(p->h_in, iq, in_bytes);
        rc = launch(p, in_kind, d_in, n_frames, flip, p->mode, d_out, p->stream, rot_delta, rot_phase0);
        if (rc) return rc;
        FSEA_HIP(hipStreamSynchronize(p->stream));
        std::memcpy(out, p->h_out, out_bytes);
        return FSEA_OK;
    }
    return exec_host_pipelined(p, in_kind, iq, 2, n_frames, flip, out, rot_delta, rot_phase0);
}

}  // namespace

extern "C" {

int fsea_exec_u8_host(fsea_plan *p, const uint8_t *iq, size_t n_frames, int flip, void *out) {
    return exec_u8_host(p, fsea::IN_U8, iq, n_frames, flip, out, 0.0, 0.0);
}

int fsea_exec_u8_shifted_device(fsea_plan *p, const void *d_iq, size_t n_frames, int flip, double cycles_per_sample,
                                double phase0_cycles, void *d_out, void *stream) {
    int rc = check_exec_args(p, d_iq, d_out, 16);
    if (rc) return rc;
    if (!std::isfinite(cycles_per_sample) || !std::isfinite(phase0_cycles)) {
        return fail(FSEA_EINVAL, "frequency shift must be finite");
    }
    FSEA_ON_DEVICE(p->device);
    return launch(p, fsea::IN_U8_ROT, d_iq, n_frames, flip, p->mode, d_out, static_cast<hipStream_t>(stream),
                  cycles_per_sample, phase0_cycles);
}

int fsea_exec_u8_shifted_host(fsea_plan *p, const uint8_t *iq, size_t n_frames, int flip, double cycles_per_sample,
                              double phase0_cycles, void *out) {
    if (!std::isfinite(cycles_per_sample) || !std::isfinite(phase0_cycles)) {
        return fail(FSEA_EINVAL, "frequency shift must be finite");
    }
    return exec_u8_host(p, fsea::IN_U8_ROT, iq, n_frames, flip, out, cycles_per_sample, phase0_cycles);
}

int fsea_exec_f64_host(fsea_plan *p, const double *iq, size_t n_frames, void *out) {
    if (!p) return fail(FSEA_EINVAL, "plan is NULL");
    if (n_frames == 0) return FSEA_OK;
    if (!iq || !out) return fail(FSEA_EINVAL, "NULL buffer");
    std::lock_guard<std::mutex> lock(p->mu);
    FSEA_ON_DEVICE(p->device);
    const size_t n_samples = (n_frames - 1) * (size_t)p->hop + (size_t)p->n;
    const size_t out_bytes = n_frames * fsea_plan_row_bytes(p);
    if (n_samples * 2 * sizeof(float) + out_bytes <= FSEA_ZERO_COPY_MAX) {
        // small batch (nrf_fft_process on a shifter buffer): narrow to f32 on the host, straight into
        // the mapped staging; one launch, one synchronisation
        int rc0 = ensure_pinned(&p->h_in, &p->h_in_bytes, n_samples * 2 * sizeof(float));
        if (rc0) return rc0;
        rc0 = ensure_pinned(&p->h_out, &p->h_out_bytes, out_bytes);
        if (rc0) return rc0;
        void *d_in = nullptr, *d_out = nullptr;
        FSEA_HIP(hipHostGetDevicePointer(&d_in, p->h_in, 0));
        FSEA_HIP(hipHostGetDevicePointer(&d_out, p->h_out, 0));
        float *dst = static_cast<float *>(p->h_in);
        for (size_t i = 0; i < n_samples * 2; ++i) dst[i] = (float)iq[i];
        rc0 = launch(p, fsea::IN_F32, d_in, n_frames, 0, p->mode, d_out, p->stream);
        if (rc0) return rc0;
        FSEA_HIP(hipStreamSynchronize(p->stream));
        std::memcpy(out, p->h_out, out_bytes);
        return FSEA_OK;
    }
    return exec_host_pipelined(p, fsea::IN_F32, iq, 16, n_frames, 0, out, 0.0, 0.0);
}

// ---- device-resident history ring (nrf_fft's history behind NRF_FFT_HISTORY=device) ----
}  // extern "C"

struct fsea_history {
    fsea_plan *plan = nullptr;
    int device = 0;           // the plan's device (fsea_history_destroy must not need the plan any more)
    int rows = 0;
    int head = 0;             // ring row holding the newest spectrum
    int cur = 0;              // which of the two storages is live (nrf_fft_shift works out of place)
    float *d_ring[2] = {nullptr, nullptr};
    float *h_stage = nullptr; // pinned, rows * n floats: target of the D2H in fsea_history_get_f64
};

namespace {

// one frame from host memory, output row written straight into device memory
int push_frame(fsea_history *h, int in_kind, const void *host_in, size_t in_bytes, int flip) {
    fsea_plan *p = h->plan;
    std::lock_guard<std::mutex> lock(p->mu);
    FSEA_ON_DEVICE(p->device);
    int rc = ensure_pinned(&p->h_in, &p->h_in_bytes, in_bytes);
    if (rc) return rc;
    void *d_in = nullptr;
    FSEA_HIP(hipHostGetDevicePointer(&d_in, p->h_in, 0));
    if (in_kind == fsea::IN_F32) {
        const double *src = static_cast<const double *>(host_in);
        float *dst = static_cast<float *>(p->h_in);
        for (size_t i = 0; i < in_bytes / sizeof(float); ++i) dst[i] = (float)src[i];
    } else {
        std::memcpy(p->h_in, host_in, in_bytes);
    }
    const int new_head = (h->head + h->rows - 1) % h->rows;  // the ring head moves back by one
    float *row = h->d_ring[h->cur] + (size_t)new_head * (size_t)p->n;
    rc = launch(p, in_kind, d_in, 1, flip, FSEA_MODE_MAG_F32, row, p->stream);
    if (rc) return rc;
    FSEA_HIP(hipStreamSynchronize(p->stream));  // the staging is free again, the row is in place
    h->head = new_head;
    return FSEA_OK;
}

}  // namespace

extern "C" {

int fsea_history_create(fsea_plan *p, int rows, fsea_history **out) {
    if (!out) return fail(FSEA_EINVAL, "history out-pointer is NULL");
    *out = nullptr;
    if (!p || rows <= 0) return fail(FSEA_EINVAL, "history needs a plan and a positive row count");
    if (p->mode != FSEA_MODE_MAG_F32) return fail(FSEA_EINVAL, "a history holds MAG_F32 rows");
    FSEA_ON_DEVICE(p->device);
    fsea_history *h = new (std::nothrow) fsea_history();
    if (!h) return fail(FSEA_ENOMEM, "out of host memory");
    h->plan = p;
    h->device = p->device;
    h->rows = rows;
    const size_t bytes = (size_t)rows * (size_t)p->n * sizeof(float);
    hipError_t he = hipMalloc(reinterpret_cast<void **>(&h->d_ring[0]), bytes);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void **>(&h->d_ring[1]), bytes);
    if (he == hipSuccess) he = hipMemset(h->d_ring[0], 0, bytes);
    if (he == hipSuccess) he = hipHostMalloc(reinterpret_cast<void **>(&h->h_stage), bytes, hipHostMallocDefault);
    if (he == hipSuccess) he = hipDeviceSynchronize();
    if (he != hipSuccess) {
        int rc = fail(FSEA_EHIP, "history setup failed: %s", hipGetErrorString(he));
        fsea_history_destroy(h);
        return rc;
    }
    *out = h;
    return FSEA_OK;
}

int fsea_history_destroy(fsea_history *h) {
    if (!h) return FSEA_OK;
    DeviceGuard device_guard_(h->device);
    if (h->d_ring[0]) (void)hipFree(h->d_ring[0]);
    if (h->d_ring[1]) (void)hipFree(h->d_ring[1]);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    delete h;
    return FSEA_OK;
}

int fsea_history_push_u8_host(fsea_history *h, const uint8_t *iq, int flip) {
    if (!h || !iq) return fail(FSEA_EINVAL, "NULL argument");
    return push_frame(h, fsea::IN_U8, iq, 2 * (size_t)h->plan->n, flip);
}

int fsea_history_push_f64_host(fsea_history *h, const double *iq) {
    if (!h || !iq) return fail(FSEA_EINVAL, "NULL argument");
    return push_frame(h, fsea::IN_F32, iq, 2 * sizeof(float) * (size_t)h->plan->n, 0);
}

int fsea_history_shift(fsea_history *h, int shift) {
    if (!h) return fail(FSEA_EINVAL, "history is NULL");
    if (shift == 0) return FSEA_OK;
    fsea_plan *p = h->plan;
    std::lock_guard<std::mutex> lock(p->mu);
    FSEA_ON_DEVICE(p->device);
    const size_t total = (size_t)h->rows * (size_t)p->n;
    if (shift >= p->n || shift <= -p->n) {  // shifted out of range: start over (src/nrf.c:574-576)
        FSEA_HIP(hipMemsetAsync(h->d_ring[h->cur], 0, total * sizeof(float), p->stream));
    } else {
        unsigned blocks = (unsigned)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(fsea_history_shift_kernel, dim3(blocks), dim3(256), 0, p->stream, h->d_ring[h->cur],
                           h->d_ring[h->cur ^ 1], p->n, total, shift);
        FSEA_HIP(hipGetLastError());
        h->cur ^= 1;
    }
    FSEA_HIP(hipStreamSynchronize(p->stream));
    return FSEA_OK;
}

int fsea_history_get_f64(fsea_history *h, double *out) {
    if (!h || !out) return fail(FSEA_EINVAL, "NULL argument");
    fsea_plan *p = h->plan;
    std::lock_guard<std::mutex> lock(p->mu);
    FSEA_ON_DEVICE(p->device);
    const size_t n = (size_t)p->n;
    const size_t first = (size_t)(h->rows - h->head);  // rows from the head to the end of storage
    const float *ring = h->d_ring[h->cur];
    // one device-to-host transfer of rows * n f32, already in newest-first order, then one widening
    FSEA_HIP(hipMemcpyAsync(h->h_stage, ring + (size_t)h->head * n, first * n * sizeof(float), hipMemcpyDeviceToHost,
                            p->stream));
    if (h->head > 0) {
        FSEA_HIP(hipMemcpyAsync(h->h_stage + first * n, ring, (size_t)h->head * n * sizeof(float), hipMemcpyDeviceToHost,
                                p->stream));
    }
    FSEA_HIP(hipStreamSynchronize(p->stream));
    const size_t total = (size_t)h->rows * n;
    const float *src = h->h_stage;
    for (size_t i = 0; i < total; ++i) out[i] = (double)src[i];
    return FSEA_OK;
}

int fsea_mean_magnitude_u8_device(fsea_plan *p, const void *d_iq, size_t n_frames, int flip, double *mean,
                                  void *stream) {
    if (!p || !mean) return fail(FSEA_EINVAL, "NULL argument");
    if (!d_iq) return fail(FSEA_EINVAL, "NULL buffer");
    if (n_frames == 0) {
        *mean = 0.0;
        return FSEA_OK;
    }
    std::lock_guard<std::mutex> lock(p->mu);
    FSEA_ON_DEVICE(p->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t count = n_frames * (size_t)p->n;
    int rc = ensure(&p->d_aux, &p->d_aux_bytes, count * sizeof(float));
    if (rc) return rc;
    rc = launch(p, fsea::IN_U8, d_iq, n_frames, flip, FSEA_MODE_MAG_NODC_F32, p->d_aux, s);
    if (rc) return rc;
    FSEA_HIP(hipMemsetAsync(p->d_acc, 0, sizeof(double), s));
    unsigned blocks = (unsigned)((count + 1023) / 1024);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(fsea_sum_f32_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const float *>(p->d_aux), count,
                       p->d_acc);
    double total = 0.0;
    FSEA_HIP(hipMemcpyAsync(&total, p->d_acc, sizeof(double), hipMemcpyDeviceToHost, s));
    FSEA_HIP(hipStreamSynchronize(s));
    *mean = total / (double)count;
    return FSEA_OK;
}

int fsea_composite_max_device(void *d_dst, const void *d_src, uint32_t dst_x, uint32_t dst_y, uint32_t width,
                              uint32_t height, uint32_t dst_stride, uint32_t dst_height, uint32_t src_stride, int device,
                              void *stream) {
    if (!d_dst || !d_src) return fail(FSEA_EINVAL, "NULL buffer");
    if (width == 0 || height == 0) return FSEA_OK;
    if ((uint64_t)dst_x + (uint64_t)width > (uint64_t)dst_stride || width > src_stride) {
        return fail(FSEA_EINVAL, "tile does not fit the row stride");
    }
    if ((uint64_t)dst_y + (uint64_t)height > (uint64_t)dst_height) {
        return fail(FSEA_EINVAL, "tile rows %u..%llu do not fit the %u destination rows", dst_y,
                    (unsigned long long)dst_y + height, dst_height);
    }
    FSEA_ON_DEVICE(device);
    const unsigned bx = 64;
    dim3 grid((width / 4 + bx) / bx, height < 4096 ? height : 4096);
    hipLaunchKernelGGL(fsea_composite_max_kernel, grid, dim3(bx), 0, static_cast<hipStream_t>(stream),
                       static_cast<uint8_t *>(d_dst), static_cast<const uint8_t *>(d_src), dst_x, dst_y, width, height,
                       dst_stride, src_stride);
    FSEA_HIP(hipGetLastError());
    return FSEA_OK;
}

int fsea_stitch_tiles_device(void *d_image, const void *d_tiles, uint32_t n_tiles, uint32_t first_x,
                             uint32_t width_step, uint32_t width, uint32_t height, uint32_t image_stride, int device,
                             void *stream) {
    if (!d_image || !d_tiles) return fail(FSEA_EINVAL, "NULL buffer");
    if (n_tiles == 0 || width == 0 || height == 0) return FSEA_OK;
    if (width_step == 0) return fail(FSEA_EINVAL, "width_step must be positive");
    if ((size_t)first_x + (size_t)(n_tiles - 1) * width_step + width > image_stride) {
        return fail(FSEA_EINVAL, "tiles do not fit the image row stride");
    }
    FSEA_ON_DEVICE(device);
    // tiles k and k + m overlap when m * step < width: m phases keep every launch race-free
    const uint32_t phases = (width + width_step - 1) / width_step;
    const unsigned bx = 64;
    for (uint32_t ph = 0; ph < phases; ++ph) {
        const uint32_t cnt = (n_tiles > ph) ? (n_tiles - ph + phases - 1) / phases : 0;
        if (cnt == 0) continue;
        dim3 grid((width / 4 + bx) / bx, height < 1024 ? height : 1024, cnt);
        hipLaunchKernelGGL(fsea_stitch_tiles_kernel, grid, dim3(bx), 0, static_cast<hipStream_t>(stream),
                           static_cast<uint8_t *>(d_image), static_cast<const uint8_t *>(d_tiles), n_tiles, first_x,
                           width_step, width, height, image_stride, phases, ph);
    }
    FSEA_HIP(hipGetLastError());
    return FSEA_OK;
}

int fsea_device_alloc(int device, size_t bytes, void **d_ptr) {
    if (!d_ptr) return fail(FSEA_EINVAL, "NULL out-pointer");
    FSEA_ON_DEVICE(device);
    FSEA_HIP(hipMalloc(d_ptr, bytes ? bytes : 16));
    return FSEA_OK;
}

int fsea_device_free(int device, void *d_ptr) {
    if (!d_ptr) return FSEA_OK;
    FSEA_ON_DEVICE(device);
    FSEA_HIP(hipFree(d_ptr));
    return FSEA_OK;
}

int fsea_host_alloc(size_t bytes, void **ptr) {
    if (!ptr) return fail(FSEA_EINVAL, "NULL out-pointer");
    *ptr = nullptr;
    FSEA_HIP(hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocPortable));
    return FSEA_OK;
}

int fsea_host_free(void *ptr) {
    if (!ptr) return FSEA_OK;
    FSEA_HIP(hipHostFree(ptr));
    return FSEA_OK;
}

int fsea_copy_to_device(int device, void *d_dst, const void *src, size_t bytes) {
    FSEA_ON_DEVICE(device);
    FSEA_HIP(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return FSEA_OK;
}

int fsea_copy_to_host(int device, void *dst, const void *d_src, size_t bytes) {
    FSEA_ON_DEVICE(device);
    FSEA_HIP(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return FSEA_OK;
}

int fsea_stream_create(int device, void **stream) {
    if (!stream) return fail(FSEA_EINVAL, "stream out-pointer is NULL");
    *stream = nullptr;
    FSEA_ON_DEVICE(device);
    hipStream_t s = nullptr;
    FSEA_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return FSEA_OK;
}

int fsea_stream_destroy(int device, void *stream) {
    if (!stream) return FSEA_OK;
    FSEA_ON_DEVICE(device);
    FSEA_HIP(hipStreamDestroy(static_cast<hipStream_t>(stream)));
    return FSEA_OK;
}

int fsea_copy_to_device_async(int device, void *d_dst, const void *src, size_t bytes, void *stream) {
    FSEA_ON_DEVICE(device);
    FSEA_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    return FSEA_OK;
}

int fsea_copy_to_host_async(int device, void *dst, const void *d_src, size_t bytes, void *stream) {
    FSEA_ON_DEVICE(device);
    FSEA_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    return FSEA_OK;
}

int fsea_stream_synchronize(fsea_plan *p, void *stream) {
    if (!p) return fail(FSEA_EINVAL, "plan is NULL");
    FSEA_ON_DEVICE(p->device);
    FSEA_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return FSEA_OK;
}

#ifdef FSEA_TUNE
}  // extern "C"

// A plain stream with the headline kernel's byte mix, 1 read : 2 written, and its cache policy (nt both ways): every thread
// moves 16 bytes in and 32 bytes out per iteration.  What the memory system gives a kernel that does nothing else.
typedef uint32_t tune_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void fsea_tune_stream_1to2_kernel(const tune_u32x4 *in, tune_u32x4 *out, size_t n16) {
    // four independent 16-byte loads in flight per thread, every wave instruction 1 KiB contiguous in both directions
    // (the two output halves are two contiguous streams, as the rows of two frames would be)
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        tune_u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __builtin_nontemporal_load(in + i + k * stride);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tune_u32x4 w = v[k];
            w.x ^= 0x80808080u;
            __builtin_nontemporal_store(v[k], out + i + k * stride);
            __builtin_nontemporal_store(w, out + n16 + i + k * stride);
        }
    }
    for (; i < n16; i += stride) {
        const tune_u32x4 v = __builtin_nontemporal_load(in + i);
        tune_u32x4 w = v;
        w.x ^= 0x80808080u;
        __builtin_nontemporal_store(v, out + i);
        __builtin_nontemporal_store(w, out + n16 + i);
    }
}

extern "C" {
// ---- tuning / measurement entry points (include/fsea_tune.h; libfsea_hip_tune.so only) ----
int fsea_tune_stream_1to2(void *const *d_in, void *const *d_out, int n_sets, size_t in_bytes, int device, void *stream,
                          int reps, float *avg_ms) {
    if (!d_in || !d_out || n_sets <= 0 || reps <= 0 || !avg_ms || (in_bytes % 16) != 0) return fail(FSEA_EINVAL, "bad arguments");
    FSEA_ON_DEVICE(device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipDeviceProp_t prop;
    FSEA_HIP(hipGetDeviceProperties(&prop, device));
    hipEvent_t e0, e1;
    FSEA_HIP(hipEventCreate(&e0));
    FSEA_HIP(hipEventCreate(&e1));
    const char *wg_env = std::getenv("FSEA_TUNE_COPY_WG");  // workgroups per CU (default 32: two 16-byte pieces per thread; measured 0.58-0.67 of 8 TB/s at 2-16, 0.73 at 32)
    const unsigned grid = (unsigned)prop.multiProcessorCount * (unsigned)(wg_env ? std::atoi(wg_env) : 32);
    FSEA_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) {
        hipLaunchKernelGGL(fsea_tune_stream_1to2_kernel, dim3(grid), dim3(256), 0, s, static_cast<const tune_u32x4 *>(d_in[i % n_sets]),
                           static_cast<tune_u32x4 *>(d_out[i % n_sets]), in_bytes / 16);
    }
    FSEA_HIP(hipEventRecord(e1, s));
    FSEA_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    FSEA_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = ms / (float)reps;
    return FSEA_OK;
}

int fsea_plan_create_variant(fsea_plan **out, int fft_size, int hop, int mode, int device, const char *variant) {
    return create_plan(out, fft_size, hop, mode, device, variant ? variant : "");
}

// Diagnostics (FSEA_TRACE=1): copies the [grid][32] trace words of the last launch.
int fsea_plan_read_trace(fsea_plan *p, unsigned long long *out, unsigned n_workgroups) {
    if (!p || !p->d_trace || n_workgroups > 4096) return fail(FSEA_EINVAL, "tracing is not enabled for this plan");
    FSEA_ON_DEVICE(p->device);
    FSEA_HIP(hipMemcpy(out, p->d_trace, (size_t)n_workgroups * 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return FSEA_OK;
}

int fsea_time_exec_u8_device(fsea_plan *p, const void *d_iq, size_t n_frames, int flip, void *d_out, void *stream,
                             int reps, float *avg_ms) {
    int rc = check_exec_args(p, d_iq, d_out, 16);
    if (rc) return rc;
    if (reps <= 0 || !avg_ms) return fail(FSEA_EINVAL, "reps must be > 0 and avg_ms non-NULL");
    FSEA_ON_DEVICE(p->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    FSEA_HIP(hipEventRecord(p->ev0, s));
    for (int i = 0; i < reps; ++i) {
        rc = launch(p, fsea::IN_U8, d_iq, n_frames, flip, p->mode, d_out, s);
        if (rc) return rc;
    }
    FSEA_HIP(hipEventRecord(p->ev1, s));
    FSEA_HIP(hipEventSynchronize(p->ev1));
    float ms = 0.f;
    FSEA_HIP(hipEventElapsedTime(&ms, p->ev0, p->ev1));
    *avg_ms = ms / (float)reps;
    return FSEA_OK;
}

// The same over n_sets independent buffer sets used in rotation, so that no launch finds its
// bytes in the 256 MiB Infinity Cache (the streaming regime bench.py measures).
int fsea_time_exec_u8_rotating(fsea_plan *p, void *const *d_iq, void *const *d_out, int n_sets, size_t n_frames,
                               int flip, void *stream, int reps, float *avg_ms) {
    if (!p || !d_iq || !d_out || n_sets <= 0 || reps <= 0 || !avg_ms) return fail(FSEA_EINVAL, "bad arguments");
    FSEA_ON_DEVICE(p->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    FSEA_HIP(hipEventRecord(p->ev0, s));
    for (int i = 0; i < reps; ++i) {
        int rc = launch(p, fsea::IN_U8, d_iq[i % n_sets], n_frames, flip, p->mode, d_out[i % n_sets], s);
        if (rc) return rc;
    }
    FSEA_HIP(hipEventRecord(p->ev1, s));
    FSEA_HIP(hipEventSynchronize(p->ev1));
    float ms = 0.f;
    FSEA_HIP(hipEventElapsedTime(&ms, p->ev0, p->ev1));
    *avg_ms = ms / (float)reps;
    return FSEA_OK;
}
#endif  // FSEA_TUNE

}  // extern "C"
