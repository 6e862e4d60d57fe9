// TEST-ONLY: runs the real kernel body (fsea_fft_core.h) on the CPU, see hip/hip_runtime.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <barrier>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "fsea_configs_tune.h"
#include "fsea_fft_core.h"
#include "fsea_tables.h"

thread_local emu_dim3 threadIdx, blockIdx, blockDim, gridDim;
static thread_local std::barrier<> *g_barrier = nullptr;
static thread_local std::barrier<> *g_wave_barrier = nullptr;
static thread_local unsigned *g_wave_slot = nullptr;  // one word per wave for readfirstlane
static thread_local unsigned *g_wave_lanes = nullptr; // 64 words per wave for the cross-lane reads
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

template <class Cfg, int IN, int MODE_T, bool ROT = false, bool RUNS = false>
static void run_grid(fsea::FftArgs a, unsigned grid) {
    if (ROT) {
        fsea::TwPair rows[32];
        fsea::build_rotation_rows(Cfg::N, Cfg::R(0), a.rot_delta, rows);
        for (int r = 0; r < 32; ++r) a.rot_row[r] = fsea::cf{rows[r].re, rows[r].im};
    }
    std::vector<fsea::TwPair> tw;
    size_t off[5];
    const int radix[4] = {Cfg::R(0), Cfg::R(1), Cfg::R(2), Cfg::R(3)};
    fsea::build_twiddles(Cfg::NP, radix, tw, off);
    for (int i = 0; i < 4; ++i) a.tw[i] = reinterpret_cast<const fsea::cf *>(tw.data()) + off[i];
    a.tw_small = reinterpret_cast<const fsea::cf *>(tw.data());
    std::vector<fsea::TwPair> twd;
    fsea::build_deferred_table(Cfg::R(0), Cfg::R(1), twd);
    twd.resize(twd.size() + 2);  // 16-byte alignment slack
    const fsea::TwPair *twd_p = twd.data();
    if (reinterpret_cast<uintptr_t>(twd_p) & 15) {
        std::memmove(twd.data() + 1, twd.data(), (twd.size() - 2) * sizeof(fsea::TwPair));
        twd_p = twd.data() + 1;
    }
    a.tw_def = reinterpret_cast<const fsea::cf *>(twd_p);
    std::vector<unsigned> ctr(9 * 32 + 2048, 0u);
    a.ctr = ctr.data();
    for (unsigned b = 0; b < grid; ++b) {
        std::vector<fsea::cf> lds_store(Cfg::LDS_ALLOC + 2);
        fsea::cf *lds = lds_store.data();
        if (reinterpret_cast<uintptr_t>(lds) & 15) lds += 1;  // 16-byte alignment as on the device
        std::barrier<> bar(Cfg::WG);
        constexpr int WAVES = (Cfg::WG + 63) / 64;
        std::vector<std::unique_ptr<std::barrier<>>> wbar;
        std::vector<unsigned> wslot(WAVES, 0);
        std::vector<unsigned> wlanes(WAVES * 64, 0);
        for (int w = 0; w < WAVES; ++w) {
            wbar.emplace_back(new std::barrier<>(std::min(64, Cfg::WG - 64 * w)));
        }
        std::vector<std::thread> th;
        for (int t = 0; t < Cfg::WG; ++t) {
            th.emplace_back([&, t] {
                threadIdx.x = (unsigned)t;
                blockIdx.x = b;
                blockDim.x = Cfg::WG;
                gridDim.x = grid;
                g_barrier = &bar;
                g_wave_barrier = wbar[t / 64].get();
                g_wave_slot = &wslot[t / 64];
                g_wave_lanes = &wlanes[(t / 64) * 64];
                fsea::FftKernel<Cfg, IN, MODE_T, ROT, RUNS>::run(a, lds);
            });
        }
        for (auto &x : th) x.join();
    }
    // the last worker of the launch must have reset the ticket counter for the next launch
    for (unsigned c : ctr) {
        if (c != 0) std::abort();
    }
}

template <class Cfg>
static int dispatch(int in_kind, int mode_t, const fsea::FftArgs &a, unsigned grid) {
    constexpr bool U8_ONLY = (Cfg::OPT & 1048576) != 0;  // the W64 schedule has u8 kernels only
    if constexpr (U8_ONLY) {
        if (in_kind != fsea::IN_U8) return -3;
    }
    if constexpr (U8_ONLY) {
        if (mode_t == fsea::MODE_MAG) run_grid<Cfg, fsea::IN_U8, fsea::MODE_MAG>(a, grid);
        else if (mode_t == fsea::MODE_DB5_U8_DCFIX) run_grid<Cfg, fsea::IN_U8, fsea::MODE_DB5_U8_DCFIX>(a, grid);
        else if (mode_t == fsea::MODE_DB10_U8) run_grid<Cfg, fsea::IN_U8, fsea::MODE_DB10_U8>(a, grid);
        else run_grid<Cfg, fsea::IN_U8, -1>(a, grid);
        return 0;
    } else
    if (in_kind == 3) {  // the half-overlap MAG kernel (K_U8_MAG_HALF): hop == N/2, runs of g_run_len frames
        if constexpr (Cfg::FPW == 1 && (Cfg::OPT & (64 | 512 | 1048576)) == 0) run_grid<Cfg, fsea::IN_U8, fsea::MODE_MAG, false, true>(a, grid);
        else return -4;
    } else
    if (in_kind == fsea::IN_U8_ROT) run_grid<Cfg, fsea::IN_U8, -1, true>(a, grid);
    else if (in_kind == fsea::IN_U8 && mode_t == fsea::MODE_MAG) run_grid<Cfg, fsea::IN_U8, fsea::MODE_MAG>(a, grid);
    else if (in_kind == fsea::IN_U8 && mode_t == fsea::MODE_DB5_U8_DCFIX) run_grid<Cfg, fsea::IN_U8, fsea::MODE_DB5_U8_DCFIX>(a, grid);
    else if (in_kind == fsea::IN_U8 && mode_t == fsea::MODE_DB10_U8) run_grid<Cfg, fsea::IN_U8, fsea::MODE_DB10_U8>(a, grid);
    else if (in_kind == fsea::IN_U8) run_grid<Cfg, fsea::IN_U8, -1>(a, grid);
    else run_grid<Cfg, fsea::IN_F32, -1>(a, grid);
    return 0;
}

// n: transform size; in_kind 0 = u8 IQ, 1 = f32 complex; specialised != 0 selects the
// compile-time MAG kernel; grid = number of emulated workgroups.
extern "C" int emu_fft_variant(int n, const char *variant, int in_kind, int specialised, const void *in, void *out,
                               size_t n_frames, size_t hop, int flip, int mode, unsigned grid);

extern "C" int emu_fft(int n, int in_kind, int specialised, const void *in, void *out, size_t n_frames, size_t hop,
                       int flip, int mode, unsigned grid) {
    return emu_fft_variant(n, "", in_kind, specialised, in, out, n_frames, hop, flip, mode, grid);
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
    if (!v.empty()) {
#define EMU_VARIANT(NN, NAME, CFG) \
    if (n == NN && v == NAME) return dispatch<fsea::FftCfg<CFG>>(in_kind, mt, a, grid);
        EMU_VARIANT(4096, "w64", FSEA_CFG_4096_W64)
        EMU_VARIANT(4096, "s2", FSEA_CFG_4096_S2)
        EMU_VARIANT(4096, "w64b", FSEA_CFG_4096_W64B)
        EMU_VARIANT(4096, "pk", FSEA_CFG_4096_PK)
        EMU_VARIANT(4096, "px0", FSEA_CFG_4096_PX0)
        EMU_VARIANT(8192, "pk", FSEA_CFG_8192_PK)
        EMU_VARIANT(8192, "px0", FSEA_CFG_8192_PX0)
        EMU_VARIANT(256, "pk", FSEA_CFG_256_PK)
        EMU_VARIANT(256, "px0", FSEA_CFG_256_PX0)
        EMU_VARIANT(1024, "px0", FSEA_CFG_1024_PX0)
        EMU_VARIANT(256, "p64", FSEA_CFG_256_P64)
        EMU_VARIANT(8192, "B2", FSEA_CFG_8192_B2)
        EMU_VARIANT(8192, "D2", FSEA_CFG_8192_D2)
        EMU_VARIANT(8192, "W", FSEA_CFG_8192_W)
        EMU_VARIANT(8192, "static", FSEA_CFG_8192_STATIC)
        EMU_VARIANT(4096, "nr", FSEA_CFG_4096_LR)
        EMU_VARIANT(2048, "nr", FSEA_CFG_2048_LR)
        EMU_VARIANT(8192, "twe", FSEA_CFG_8192_TWE)
        EMU_VARIANT(4096, "twe", FSEA_CFG_4096_TWE)
        EMU_VARIANT(1024, "twe", FSEA_CFG_1024_TWE)
        EMU_VARIANT(8192, "r1", FSEA_CFG_8192_R1)
        EMU_VARIANT(8192, "nd", FSEA_CFG_8192_ND)
        EMU_VARIANT(8192, "v2", FSEA_CFG_8192_V2)
        EMU_VARIANT(8192, "v2s", FSEA_CFG_8192_V2S)
        EMU_VARIANT(8192, "tk", FSEA_CFG_8192_TK)
        EMU_VARIANT(8192, "pr", FSEA_CFG_8192_PR)
        EMU_VARIANT(8192, "x0", FSEA_CFG_8192_X0)
        EMU_VARIANT(8192, "x7", FSEA_CFG_8192_X7)
        EMU_VARIANT(8192, "A", FSEA_CFG_8192_A)
        EMU_VARIANT(8192, "B", FSEA_CFG_8192_B)
        EMU_VARIANT(8192, "D", FSEA_CFG_8192_D)
        EMU_VARIANT(8192, "notwl", FSEA_CFG_8192_NOTWL)
        EMU_VARIANT(8192, "notwr", FSEA_CFG_8192_NOTWR)
        EMU_VARIANT(4096, "x0", FSEA_CFG_4096_X0)
        EMU_VARIANT(4096, "df", FSEA_CFG_4096_DF)
        EMU_VARIANT(4096, "B", FSEA_CFG_4096_B)
        EMU_VARIANT(256, "p16", FSEA_CFG_256_P16)
        EMU_VARIANT(128, "p16", FSEA_CFG_128_P16)
        EMU_VARIANT(4096, "f1", FSEA_CFG_4096_F1)
        EMU_VARIANT(4096, "r1", FSEA_CFG_4096_R1)
        EMU_VARIANT(4096, "t256", FSEA_CFG_4096_T256)
        EMU_VARIANT(4096, "B3", FSEA_CFG_4096_B3)
        EMU_VARIANT(4096, "C", FSEA_CFG_4096_C)
        EMU_VARIANT(4096, "D", FSEA_CFG_4096_D)
        EMU_VARIANT(2048, "x0", FSEA_CFG_2048_X0)
        EMU_VARIANT(2048, "df", FSEA_CFG_2048_DF)
        EMU_VARIANT(2048, "B", FSEA_CFG_2048_B)
        EMU_VARIANT(2048, "C", FSEA_CFG_2048_C)
        EMU_VARIANT(1024, "r1", FSEA_CFG_1024_R1)
        EMU_VARIANT(1024, "x0", FSEA_CFG_1024_X0)
        EMU_VARIANT(1024, "B", FSEA_CFG_1024_B)
        EMU_VARIANT(1024, "C", FSEA_CFG_1024_C)
        EMU_VARIANT(1024, "D", FSEA_CFG_1024_D)
        EMU_VARIANT(16384, "r1", FSEA_CFG_16384_R1)
        EMU_VARIANT(16384, "nd", FSEA_CFG_16384_ND)
        EMU_VARIANT(16384, "B", FSEA_CFG_16384_B)
        return -2;
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
