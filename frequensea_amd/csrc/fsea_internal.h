// fsea_internal.h -- what the translation units of libfsea_hip.so share behind the C ABI (include/fsea.h): the plan
// object, error plumbing, and the launch dispatcher.  fsea_api.hip: plans, the power-of-two launches, every entry point;
// fsea_anysize.hip: the transform sizes without a kernel of their own (Bluestein's algorithm, four-step decomposition).
#pragma once

#include "../../include/fsea.h"

#include <hip/hip_runtime.h>

#include <mutex>
#include <string>

#include "fsea_registry.h"

#define FSEA_STATIC_UNITS_PER_WG 16u  // measured crossover: profiles/r02_static_vs_ticket_distribution.txt
#define FSEA_CTR_SLOTS 64u          // ticket-counter slots = streams one plan may be launched on concurrently
#define FSEA_HOST_CHUNKS_MAX 16      // chunks of one host-buffer call (fsea_exec_*_host) in flight
#define FSEA_CTR_WORDS (9u * 32u + 2048u)  // 8 ticket pools + the finished-workgroups word, one 128-byte line each; one progress word per workgroup (tuning option)


namespace fsea_detail {

int fail(int code, const char *fmt, ...);
size_t mode_elem_bytes(int mode);

#define FSEA_HIP(call)                                                                            \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            return fsea_detail::fail(FSEA_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                                \
        }                                                                                         \
    } while (0)


// Every entry point works on the plan's device and leaves the caller's current device as it was
// (a host process driving several GPUs, torch included, keeps its own notion of "current").
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) err = hipSetDevice(device);
        else prev = -1;  // nothing to restore
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define FSEA_ON_DEVICE(dev)                                                                       \
    DeviceGuard device_guard_(dev);                                                               \
    if (device_guard_.err != hipSuccess) {                                                        \
        return fsea_detail::fail(FSEA_EHIP, "hipSetDevice(%d) failed: %s", (dev), hipGetErrorString(device_guard_.err)); \
    }

// where the rows of a launch go when they are tiles of an image (fsea_exec_u8_tiled_device); rows == 0: contiguous
struct TileLayout {
    uint32_t rows = 0, pitch_row = 0, pitch_tile = 0;
    size_t span = 0;
};


}  // namespace fsea_detail

struct fsea_plan {
    int n = 0;
    int hop = 0;
    int mode = 0;
    int device = 0;
    const fsea::KernelEntry *entry = nullptr;
    hipStream_t stream = nullptr;
    fsea::cf *d_tw = nullptr;      // passes 1..np-1 concatenated
    size_t tw_off[5] = {0, 0, 0, 0, 0};  // passes 0..3, then the HI/LO factor tables (fsea_tables.h)
    size_t tw_def_off = 0;               // deferred middle-pass table (fo::DEFER / fo::V2), 16-byte aligned
    int num_cu = 0;
    // Ticket counters of the multi-wave sizes: one slot per stream the plan is launched on.  Launches
    // on one stream run in order and the last workgroup of a launch zeroes its slot, so a stream
    // needs exactly one; launches on different streams may overlap and never share one.
    unsigned *d_ctr = nullptr;  // FSEA_CTR_SLOTS x FSEA_CTR_WORDS
    std::mutex slot_mu;
    struct CtrSlot {
        hipStream_t stream = nullptr;   // the stream the slot serves (meaningful while `used` and not `anonymous`)
        hipEvent_t ev = nullptr;        // recorded behind the slot's last launch (not while the stream is being captured)
        bool used = false, pending = false, anonymous = false, captured = false;
        bool launching = false;         // claimed by a host thread between counter_slot() and the record of `ev`: not to be recycled
        unsigned long long seq = 0;     // launch order, for least-recently-used recycling
    } slots[FSEA_CTR_SLOTS];
    unsigned long long slot_seq = 0;
    unsigned long long *d_trace = nullptr;  // FSEA_TRACE diagnostics (tuning library)
    int occ[fsea::K_COUNT] = {};
    // FSEA_UNITS_AUTO: launches with at most FSEA_STATIC_UNITS_PER_WG units per workgroup use the static interleave,
    // longer ones the ticket pools; fsea_plan_set_unit_distribution pins one of the two
    int units_policy = FSEA_UNITS_AUTO;
    int half_run_max = 32;         // frames per run of the half-overlap kernels at most (FSEA_HALF_RUN_MAX at plan creation; 8, 16 and 32 equal in rate, fetch 1.127x / 1.064x / 1.033x the distinct bytes: profiles/r04_stft_run_length.txt)
    bool no_half_overlap = false;  // FSEA_NO_HALF_OVERLAP=1 at plan creation: hop == N/2 runs the ordinary kernel (A/B measurements)
    // staging for the host-buffer entry points
    std::mutex mu;
    void *d_in = nullptr;
    size_t d_in_bytes = 0;
    void *d_out = nullptr;
    size_t d_out_bytes = 0;
    void *d_aux = nullptr;
    size_t d_aux_bytes = 0;
    double *d_acc = nullptr;
    // small host batches (the nrf_fft_process pattern: one 2 KiB frame in, one row out) go through
    // pinned, device-mapped staging: the kernel reads and writes host memory itself, so a call is
    // one launch and one synchronisation instead of copy + launch + copy
    void *h_in = nullptr;
    void *h_out = nullptr;
    size_t h_in_bytes = 0, h_out_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // the pipelined host-buffer path (exec_host_pipelined): copy-in and copy-out streams beside `stream`, and one
    // "chunk arrived" / "chunk transformed" event pair per chunk in flight
    // Bluestein plans (transform sizes without a kernel of their own): `n` is the logical size, `entry` the power-of-two
    // kernel set of size blu_m the convolution runs on
    int blu_m = 0;
    fsea_plan *blu_inner = nullptr;          // size blu_m, COMPLEX_F32
    fsea::cf *d_blu_chirp = nullptr;         // conj(w[j]), j < n
    fsea::cf *d_blu_bfft = nullptr;          // FFT_m of the wrapped chirp
    fsea::cf *d_blu_dc = nullptr;            // spectrum of the offset-binary DC term, n entries
    fsea::cf *d_blu_work[2] = {nullptr, nullptr};
    size_t blu_work_frames = 0;
    // the work buffers are the plan's, not the launch's: launches of such a plan on different streams are put in order
    // behind one another (an event recorded behind each launch, waited for by the next one's stream)
    std::mutex work_mu;
    hipEvent_t work_ev = nullptr;
    bool work_pending = false;
    // four-step plans (powers of two above 16384): n = fs_n1 * fs_n2, two inner plans, the twiddles W_n^{j2 k1}
    int fs_n1 = 0, fs_n2 = 0;
    fsea_plan *fs_inner1 = nullptr, *fs_inner2 = nullptr;
    fsea::cf *d_fs_tw = nullptr;
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    hipEvent_t ev_in[FSEA_HOST_CHUNKS_MAX] = {}, ev_done[FSEA_HOST_CHUNKS_MAX] = {};
    // taper window (fsea_plan_set_window): weights in the pass-0 lane order with (-1)^n folded in, and the DC term's
    // spectrum around bin n/2 (FftArgs::win, win_dc); window_form 0 = none, 1 = centred, 2 = offset-binary
    float *d_win = nullptr;
    fsea::cf *d_win_dc = nullptr;
    int window_form = 0;
    // both strings live as long as the plan: fsea_plan_kernel_name hands out a pointer into the one in use, and a pointer a
    // caller got earlier stays valid across fsea_plan_set_window (the windowed name depends on the plan's constants only)
    std::string kernel_name;        // while no window is set
    std::string kernel_name_win;    // while one is (filled by the first fsea_plan_set_window)
};


namespace fsea_detail {

// One batch of frames through whatever serves the plan's size: a power-of-two kernel, Bluestein's algorithm, or the
// four-step decomposition (the latter two call back into this for their inner transforms).
int launch(fsea_plan *p, int in_kind, const void *d_in, size_t n_frames, int flip, int mode, void *d_out, hipStream_t s,
           double rot_delta = 0.0, double rot_phase0 = 0.0, const TileLayout *tiles = nullptr);

// fsea_anysize.hip
#define FSEA_MAX_FFT_SIZE (1 << 20)  // largest transform: four-step up to 2^20 points; Bluestein's m = 2^p >= 2n - 1 within it
bool fourstep_split(int n, int *n1, int *n2);
int bluestein_m(int n);
int blu_setup(fsea_plan *p);
int fs_setup(fsea_plan *p);
int blu_launch(fsea_plan *p, int in_kind, const void *d_in, size_t n_frames, int flip, int mode, void *d_out, hipStream_t s);
int fs_launch(fsea_plan *p, int in_kind, const void *d_in, size_t n_frames, int flip, int mode, void *d_out, hipStream_t s);

}  // namespace fsea_detail
