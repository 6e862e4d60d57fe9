// TEST-ONLY: runs the real kernel body (fsea_fft_core.h) on the CPU, see hip/hip_runtime.h.
#include "emu_common.h"

thread_local emu_dim3 threadIdx, blockIdx, blockDim, gridDim;
thread_local std::barrier<> *g_barrier = nullptr;
thread_local std::barrier<> *g_wave_barrier = nullptr;
thread_local unsigned *g_wave_slot = nullptr;  // one word per wave for readfirstlane
thread_local unsigned *g_wave_lanes = nullptr; // 64 words per wave for the cross-lane reads
void emu_syncthreads() { g_barrier->arrive_and_wait(); }
void emu_wave_barrier() { g_wave_barrier->arrive_and_wait(); }
unsigned emu_readfirstlane(unsigned v) {
    if ((threadIdx.x & 63) == 0) *g_wave_slot = v;
    g_wave_barrier->arrive_and_wait();
    const unsigned r = *g_wave_slot;
    g_wave_barrier->arrive_and_wait();
    return r;
}

unsigned fsea::emu_lane_read(unsigned v, int src_lane) {
    g_wave_lanes[threadIdx.x & 63] = v;
    g_wave_barrier->arrive_and_wait();
    const unsigned r = g_wave_lanes[src_lane & 63];
    g_wave_barrier->arrive_and_wait();
    return r;
}

// n: transform size; in_kind 0 = u8 IQ, 1 = f32 complex; specialised != 0 selects the
// compile-time MAG kernel; grid = number of emulated workgroups.
extern "C" int emu_fft_variant(int n, const char *variant, int in_kind, int specialised, const void *in, void *out,
                               size_t n_frames, size_t hop, int flip, int mode, unsigned grid);

extern "C" int emu_fft(int n, int in_kind, int specialised, const void *in, void *out, size_t n_frames, size_t hop,
                       int flip, int mode, unsigned grid) {
    return emu_fft_variant(n, "", in_kind, specialised, in, out, n_frames, hop, flip, mode, grid);
}

std::vector<float> g_window;
int g_window_mode = 2, g_window_form = 0;
// taper window of the following calls: n weights (n == 0 or w == NULL: none); mode = FftKernel's WIN (1 / 2);
// form 0 = decided from the window's spectrum as fsea_plan_set_window does, 2 = the offset-binary form forced
extern "C" void emu_set_window(const float *w, int n, int mode, int form) {
    g_window.assign(w ? w : nullptr, w ? w + n : nullptr);
    g_window_mode = mode;
    g_window_form = form;
}

static double g_rot_delta = 0.0, g_rot_phase0 = 0.0;

// in_kind 2 (frequency-shifted u8 input) takes its shift from here
extern "C" void emu_set_shift(double cycles_per_sample, double phase0_cycles) {
    g_rot_delta = cycles_per_sample;
    g_rot_phase0 = phase0_cycles;
}

// unit distribution of the multi-wave sizes (FftArgs::dynamic_units): sticky until set again
static uint32_t g_dynamic_units = 1;
extern "C" void emu_set_dynamic_units(uint32_t on) { g_dynamic_units = on; }

// half-overlap kernels (in_kind 3): frames per run
static uint32_t g_run_len = 4;
extern "C" void emu_set_run_len(uint32_t frames) { g_run_len = frames; }

// tiled output (FftArgs::tile_rows ...): set before a call, cleared by it
static uint32_t g_tile_rows = 0, g_pitch_row = 0, g_pitch_tile = 0;
static size_t g_out_span = 0;
extern "C" void emu_set_tiles(uint32_t tile_rows, uint32_t pitch_row, uint32_t pitch_tile, size_t out_span) {
    g_tile_rows = tile_rows;
    g_pitch_row = pitch_row;
    g_pitch_tile = pitch_tile;
    g_out_span = out_span;
}

extern "C" int emu_fft_variant(int n, const char *variant, int in_kind, int specialised, const void *in, void *out,
                               size_t n_frames, size_t hop, int flip, int mode, unsigned grid) {
    fsea::FftArgs a;
    a.rot_delta = g_rot_delta;
    a.rot_phase0 = g_rot_phase0;
    a.tile_rows = g_tile_rows;
    a.pitch_row = g_pitch_row;
    a.pitch_tile = g_pitch_tile;
    a.out_span = g_out_span;
    a.dynamic_units = g_dynamic_units;
    if (in_kind == 3) {
        a.run_len = g_run_len;
        a.dynamic_units = 0;
    }
    g_tile_rows = 0;
    a.in = in;
    a.out = out;
    a.n_frames = n_frames;
    a.hop = hop;
    a.xormask = flip ? 0u : 0x80808080u;
    a.mode = mode;
    a.trace = nullptr;
    // the compile-time-mode kernels (MAG, DB5, DB10) serve raw int8 input (flip)
    const bool has_fixed = mode == fsea::MODE_MAG || mode == fsea::MODE_DB5_U8_DCFIX || mode == fsea::MODE_DB10_U8;
    const int mt = (specialised && has_fixed && in_kind == fsea::IN_U8 && flip) ? mode : -1;
    const std::string v = variant ? variant : "";
    // the per-(size, mode) product configurations (fsea_configs.h)
    if (n == 256 && v == "rows") return dispatch<fsea::FftCfg<FSEA_CFG_256_ROWS>>(in_kind, mt, a, grid);
    if (n == 512 && v == "px") return dispatch<fsea::FftCfg<FSEA_CFG_512_PX>>(in_kind, mt, a, grid);
    if (n == 1024 && v == "rt") return dispatch<fsea::FftCfg<FSEA_CFG_1024_RT>>(in_kind, mt, a, grid);
    if (!v.empty()) {
        int rc = emu_variants_a(n, v, in_kind, mt, a, grid);
        if (rc == -2) rc = emu_variants_b(n, v, in_kind, mt, a, grid);
        if (rc == -2) rc = emu_variants_c(n, v, in_kind, mt, a, grid);
        return rc;
    }
    switch (n) {
    case 32: return dispatch<fsea::FftCfg<FSEA_CFG_32>>(in_kind, mt, a, grid);
    case 64: return dispatch<fsea::FftCfg<FSEA_CFG_64>>(in_kind, mt, a, grid);
    case 128: return dispatch<fsea::FftCfg<FSEA_CFG_128>>(in_kind, mt, a, grid);
    case 256: return dispatch<fsea::FftCfg<FSEA_CFG_256>>(in_kind, mt, a, grid);
    case 512: return dispatch<fsea::FftCfg<FSEA_CFG_512>>(in_kind, mt, a, grid);
    case 1024: return dispatch<fsea::FftCfg<FSEA_CFG_1024>>(in_kind, mt, a, grid);
    case 2048: return dispatch<fsea::FftCfg<FSEA_CFG_2048>>(in_kind, mt, a, grid);
    case 4096: return dispatch<fsea::FftCfg<FSEA_CFG_4096>>(in_kind, mt, a, grid);
    case 8192: return dispatch<fsea::FftCfg<FSEA_CFG_8192>>(in_kind, mt, a, grid);
    case 16384: return dispatch<fsea::FftCfg<FSEA_CFG_16384>>(in_kind, mt, a, grid);
    default: return -1;
    }
}
