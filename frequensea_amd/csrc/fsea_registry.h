// fsea_registry.h -- table of compiled kernel configurations (host side).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>

#include "fsea_fft_core.h"

namespace fsea {

// The __global__ entry points one configuration may have.  The three compile-time-mode kernels
// serve raw int8 input (flip: the bytes are the signed samples, no XOR) with the epilogue fixed:
// the nrf_fft_process path (MAG, src/nrf.c:619-630) and the sweep tools' pixel paths (DB5 with the
// DC fix, c/fft-batch-broad.c:106-121; DB10, c/fft-batch.c:83-94).  K_U8 takes any epilogue and
// either byte convention at run time (uniform branch), K_U8_ROT adds the fused frequency shift,
// K_F32 takes f32-complex input (the NUT_BUFFER_F64 branch).
// K_U8_MAG_HALF: the MAG kernel for 50 %-overlapped frames (hop == N/2) of the sizes with one frame per workgroup (8192,
// 16384): runs of consecutive frames per workgroup, every sample loaded once (FftKernel<..., RUNS = true>).
// K_U8_MAG_WIN, K_U8_WIN, K_U8_MAG_HALF_WIN, K_U8_DB5_WIN, K_U8_DB10_WIN: the u8 kernels with the taper window fused into
// pass 0 (FftKernel<..., WIN>; fsea_plan_set_window) -- what a plan with a window launches.  K_U8_ROT_WIN, K_F32_WIN: the
// frequency-shifted and the f32-complex-input kernels with the same taper (the nrf_freq_shifter -> nrf_fft chain of
// lua/fft-shifted.lua:52-55 and nrf_fft_process' F64 branch, src/nrf.c:607-612, on a plan with a window).
enum : int { K_U8_MAG = 0, K_U8_DB5 = 1, K_U8_DB10 = 2, K_U8 = 3, K_U8_ROT = 4, K_F32 = 5, K_U8_MAG_HALF = 6,
             K_U8_MAG_WIN = 7, K_U8_WIN = 8, K_U8_MAG_HALF_WIN = 9, K_U8_DB5_WIN = 10, K_U8_DB10_WIN = 11,
             K_U8_ROT_WIN = 12, K_F32_WIN = 13, K_COUNT = 14 };

struct KernelEntry {
    int n;                 // transform size
    const char *variant;   // "" = the product configuration of this n
    int t, fpw, wg, np;
    int radix[4];
    size_t lds_bytes;
    size_t lds_bytes_win;  // static LDS of the windowed compile-time MAG kernels (*_u8_mag_win, *_u8_mag_half_win); 0 = no windowed kernels
    size_t lds_bytes_win_other;  // ... of the other windowed u8 kinds (where the DC table sits in LDS it comes on top)
    int c0;                // samples per pass-0 load (hop must be a multiple)
    int counters;          // ticket counters: 0 = never used (single-wave frames), 1 = by launches with FftArgs::dynamic_units,
                           // 2 = by every launch (the V2 schedule and the progress-word experiments of the tuning library)
    const void *fn[K_COUNT];    // __global__ function addresses (occupancy queries); null = not compiled
    const char *name[K_COUNT];  // symbol names as rocprof shows them
    void (*launch)(int kind, const FftArgs &args, unsigned grid, hipStream_t stream);
    void (*launch_half)(const FftArgs &args, unsigned grid, hipStream_t stream);  // K_U8_MAG_HALF, or null
    void (*launch_win)(int kind, const FftArgs &args, unsigned grid, hipStream_t stream);  // the *_WIN kinds, or null
};

// Each k_*.hip translation unit exports `int fsea_kernels_<tag>(KernelEntry *out, int cap)`
// which fills `out` with its configurations and returns how many it has.

}  // namespace fsea

#define FSEA_KERNEL_FN_(NAME, SUFFIX, ...)                                                            \
    extern "C" __global__ __launch_bounds__(NAME##_cfg::WG, NAME##_cfg::WPE) void NAME##SUFFIX(       \
        fsea::FftArgs a) {                                                                            \
        __shared__ __attribute__((aligned(16))) fsea::cf lds[fsea::FftKernel<NAME##_cfg, __VA_ARGS__>::LDS_CF]; \
        fsea::FftKernel<NAME##_cfg, __VA_ARGS__>::run(a, lds);                                        \
    }

#define FSEA_KERNEL_ENTRY_HEAD_(NAME, VARIANT)                                                        \
    NAME##_cfg::N, VARIANT, NAME##_cfg::T, NAME##_cfg::FPW, NAME##_cfg::WG, NAME##_cfg::NP,           \
        {NAME##_cfg::R(0), NAME##_cfg::R(1), NAME##_cfg::R(2), NAME##_cfg::R(3)},                     \
        sizeof(fsea::cf) * NAME##_cfg::LDS_ALLOC, 0, 0, NAME##_cfg::C(0),                                 \
        fsea::FftKernel<NAME##_cfg, fsea::IN_U8>::counters_used()

// Defines the six __global__ entry points of one configuration, with plain C names so that
// profiles are easy to read, and the launch trampoline + KernelEntry for it.
#define FSEA_DEFINE_KERNEL(NAME, VARIANT, ...)  /* NAME: C symbol stem; VARIANT: registry key */     \
    using NAME##_cfg = fsea::FftCfg<__VA_ARGS__>;                                                     \
    FSEA_KERNEL_FN_(NAME, _u8_mag, fsea::IN_U8, fsea::MODE_MAG)                                       \
    FSEA_KERNEL_FN_(NAME, _u8_db5, fsea::IN_U8, fsea::MODE_DB5_U8_DCFIX)                              \
    FSEA_KERNEL_FN_(NAME, _u8_db10, fsea::IN_U8, fsea::MODE_DB10_U8)                                  \
    FSEA_KERNEL_FN_(NAME, _u8, fsea::IN_U8)                                                           \
    FSEA_KERNEL_FN_(NAME, _u8_rot, fsea::IN_U8, -1, true)                                             \
    FSEA_KERNEL_FN_(NAME, _f32, fsea::IN_F32)                                                         \
    static void NAME##_launch(int kind, const fsea::FftArgs &a, unsigned grid, hipStream_t s) {       \
        const dim3 g(grid), b(NAME##_cfg::WG);                                                        \
        switch (kind) {                                                                               \
        case fsea::K_U8_MAG: hipLaunchKernelGGL(NAME##_u8_mag, g, b, 0, s, a); break;                 \
        case fsea::K_U8_DB5: hipLaunchKernelGGL(NAME##_u8_db5, g, b, 0, s, a); break;                 \
        case fsea::K_U8_DB10: hipLaunchKernelGGL(NAME##_u8_db10, g, b, 0, s, a); break;               \
        case fsea::K_U8_ROT: hipLaunchKernelGGL(NAME##_u8_rot, g, b, 0, s, a); break;                 \
        case fsea::K_F32: hipLaunchKernelGGL(NAME##_f32, g, b, 0, s, a); break;                       \
        default: hipLaunchKernelGGL(NAME##_u8, g, b, 0, s, a); break;                                 \
        }                                                                                             \
    }                                                                                                 \
    static fsea::KernelEntry NAME##_entry() {                                                         \
        return fsea::KernelEntry{                                                                     \
            FSEA_KERNEL_ENTRY_HEAD_(NAME, VARIANT),                                                   \
            {reinterpret_cast<const void *>(&NAME##_u8_mag), reinterpret_cast<const void *>(&NAME##_u8_db5), \
             reinterpret_cast<const void *>(&NAME##_u8_db10), reinterpret_cast<const void *>(&NAME##_u8),    \
             reinterpret_cast<const void *>(&NAME##_u8_rot), reinterpret_cast<const void *>(&NAME##_f32)},   \
            {#NAME "_u8_mag", #NAME "_u8_db5", #NAME "_u8_db10", #NAME "_u8", #NAME "_u8_rot", #NAME "_f32"}, \
            &NAME##_launch};                                                                          \
    }

// Tuning variants (libfsea_hip_tune.so only): the MAG kernel and the run-time-mode kernel.  The pixel modes of such a
// plan run the run-time-mode kernel; its f32-input and frequency-shifted kinds do not exist (launch() fails with
// FSEA_EINVAL).  The V2 schedule (opt::V2) always hands its frames out by the ticket pools: fsea_plan_set_unit_distribution
// has no effect on a V2 variant.
#define FSEA_DEFINE_KERNEL_LITE(NAME, VARIANT, ...)                                                   \
    using NAME##_cfg = fsea::FftCfg<__VA_ARGS__>;                                                     \
    FSEA_KERNEL_FN_(NAME, _u8_mag, fsea::IN_U8, fsea::MODE_MAG)                                       \
    FSEA_KERNEL_FN_(NAME, _u8, fsea::IN_U8)                                                           \
    static void NAME##_launch(int kind, const fsea::FftArgs &a, unsigned grid, hipStream_t s) {       \
        const dim3 g(grid), b(NAME##_cfg::WG);                                                        \
        if (kind == fsea::K_U8_MAG) hipLaunchKernelGGL(NAME##_u8_mag, g, b, 0, s, a);                 \
        else hipLaunchKernelGGL(NAME##_u8, g, b, 0, s, a);                                            \
    }                                                                                                 \
    static fsea::KernelEntry NAME##_entry() {                                                         \
        return fsea::KernelEntry{                                                                     \
            FSEA_KERNEL_ENTRY_HEAD_(NAME, VARIANT),                                                   \
            {reinterpret_cast<const void *>(&NAME##_u8_mag), nullptr, nullptr,                        \
             reinterpret_cast<const void *>(&NAME##_u8), nullptr, nullptr},                           \
            {#NAME "_u8_mag", "", "", #NAME "_u8", "", ""},                                           \
            &NAME##_launch};                                                                          \
    }

// The u8 kernels only (MAG, DB5, DB10 with the mode fixed, and the run-time-mode kernel): configurations that have no
// f32-input or frequency-shifted form (the single-wave 64 x 64 schedule).
#define FSEA_DEFINE_KERNEL_U8(NAME, VARIANT, ...)                                                     \
    using NAME##_cfg = fsea::FftCfg<__VA_ARGS__>;                                                     \
    FSEA_KERNEL_FN_(NAME, _u8_mag, fsea::IN_U8, fsea::MODE_MAG)                                       \
    FSEA_KERNEL_FN_(NAME, _u8_db5, fsea::IN_U8, fsea::MODE_DB5_U8_DCFIX)                              \
    FSEA_KERNEL_FN_(NAME, _u8_db10, fsea::IN_U8, fsea::MODE_DB10_U8)                                  \
    FSEA_KERNEL_FN_(NAME, _u8, fsea::IN_U8)                                                           \
    static void NAME##_launch(int kind, const fsea::FftArgs &a, unsigned grid, hipStream_t s) {       \
        const dim3 g(grid), b(NAME##_cfg::WG);                                                        \
        switch (kind) {                                                                               \
        case fsea::K_U8_MAG: hipLaunchKernelGGL(NAME##_u8_mag, g, b, 0, s, a); break;                 \
        case fsea::K_U8_DB5: hipLaunchKernelGGL(NAME##_u8_db5, g, b, 0, s, a); break;                 \
        case fsea::K_U8_DB10: hipLaunchKernelGGL(NAME##_u8_db10, g, b, 0, s, a); break;               \
        default: hipLaunchKernelGGL(NAME##_u8, g, b, 0, s, a); break;                                 \
        }                                                                                             \
    }                                                                                                 \
    static fsea::KernelEntry NAME##_entry() {                                                         \
        return fsea::KernelEntry{                                                                     \
            FSEA_KERNEL_ENTRY_HEAD_(NAME, VARIANT),                                                   \
            {reinterpret_cast<const void *>(&NAME##_u8_mag), reinterpret_cast<const void *>(&NAME##_u8_db5), \
             reinterpret_cast<const void *>(&NAME##_u8_db10), reinterpret_cast<const void *>(&NAME##_u8),    \
             nullptr, nullptr},                                                                       \
            {#NAME "_u8_mag", #NAME "_u8_db5", #NAME "_u8_db10", #NAME "_u8", "", ""},                \
            &NAME##_launch};                                                                          \
    }

// The half-overlap MAG kernel of a configuration defined above (one frame per workgroup), and the entry that carries it.
#define FSEA_DEFINE_HALF_OVERLAP(NAME)                                                                \
    FSEA_KERNEL_FN_(NAME, _u8_mag_half, fsea::IN_U8, fsea::MODE_MAG, false, true)                     \
    static void NAME##_launch_half(const fsea::FftArgs &a, unsigned grid, hipStream_t s) {            \
        hipLaunchKernelGGL(NAME##_u8_mag_half, dim3(grid), dim3(NAME##_cfg::WG), 0, s, a);            \
    }                                                                                                 \
    static fsea::KernelEntry NAME##_entry_half() {                                                    \
        fsea::KernelEntry e = NAME##_entry();                                                         \
        e.fn[fsea::K_U8_MAG_HALF] = reinterpret_cast<const void *>(&NAME##_u8_mag_half);              \
        e.name[fsea::K_U8_MAG_HALF] = #NAME "_u8_mag_half";                                           \
        e.launch_half = &NAME##_launch_half;                                                          \
        return e;                                                                                     \
    }
#define FSEA_REGISTER_HALF(NAME) if (n < cap) out[n++] = NAME##_entry_half();

// The windowed kernels of a configuration defined above: NAME_u8_mag_win, NAME_u8_db5_win, NAME_u8_db10_win (epilogue and
// byte convention fixed at compile time, as their un-windowed twins), NAME_u8_win (and NAME_u8_mag_half_win with HALF = 1).  WMODE: FftKernel's WIN (1 = weights fetched per frame, 2 = register-resident).  FSEA_REGISTER_WIN /
// FSEA_REGISTER_HALF_WIN register the entry with them.
#define FSEA_DEFINE_WINDOWED(NAME, WMODE)                                                             \
    FSEA_KERNEL_FN_(NAME, _u8_mag_win, fsea::IN_U8, fsea::MODE_MAG, false, false, WMODE)              \
    FSEA_KERNEL_FN_(NAME, _u8_db5_win, fsea::IN_U8, fsea::MODE_DB5_U8_DCFIX, false, false, WMODE)     \
    FSEA_KERNEL_FN_(NAME, _u8_db10_win, fsea::IN_U8, fsea::MODE_DB10_U8, false, false, WMODE)         \
    FSEA_KERNEL_FN_(NAME, _u8_win, fsea::IN_U8, -1, false, false, WMODE)                              \
    FSEA_KERNEL_FN_(NAME, _u8_rot_win, fsea::IN_U8, -1, true, false, WMODE)                           \
    FSEA_KERNEL_FN_(NAME, _f32_win, fsea::IN_F32, -1, false, false, WMODE)                            \
    static void NAME##_launch_win(int kind, const fsea::FftArgs &a, unsigned grid, hipStream_t s) {   \
        const dim3 g(grid), b(NAME##_cfg::WG);                                                        \
        if (kind == fsea::K_U8_MAG_WIN) hipLaunchKernelGGL(NAME##_u8_mag_win, g, b, 0, s, a);         \
        else if (kind == fsea::K_U8_DB5_WIN) hipLaunchKernelGGL(NAME##_u8_db5_win, g, b, 0, s, a);    \
        else if (kind == fsea::K_U8_DB10_WIN) hipLaunchKernelGGL(NAME##_u8_db10_win, g, b, 0, s, a);  \
        else if (kind == fsea::K_U8_ROT_WIN) hipLaunchKernelGGL(NAME##_u8_rot_win, g, b, 0, s, a);    \
        else if (kind == fsea::K_F32_WIN) hipLaunchKernelGGL(NAME##_f32_win, g, b, 0, s, a);          \
        else hipLaunchKernelGGL(NAME##_u8_win, g, b, 0, s, a);                                        \
    }                                                                                                 \
    static void NAME##_add_win(fsea::KernelEntry &e) {                                                \
        e.fn[fsea::K_U8_ROT_WIN] = reinterpret_cast<const void *>(&NAME##_u8_rot_win);                \
        e.fn[fsea::K_F32_WIN] = reinterpret_cast<const void *>(&NAME##_f32_win);                      \
        e.name[fsea::K_U8_ROT_WIN] = #NAME "_u8_rot_win";                                             \
        e.name[fsea::K_F32_WIN] = #NAME "_f32_win";                                                   \
        e.lds_bytes_win = sizeof(fsea::cf) * fsea::FftKernel<NAME##_cfg, fsea::IN_U8, fsea::MODE_MAG, false, false, WMODE>::LDS_CF; \
        e.lds_bytes_win_other = sizeof(fsea::cf) * fsea::FftKernel<NAME##_cfg, fsea::IN_U8, -1, false, false, WMODE>::LDS_CF; \
        e.fn[fsea::K_U8_MAG_WIN] = reinterpret_cast<const void *>(&NAME##_u8_mag_win);                \
        e.fn[fsea::K_U8_DB5_WIN] = reinterpret_cast<const void *>(&NAME##_u8_db5_win);                \
        e.fn[fsea::K_U8_DB10_WIN] = reinterpret_cast<const void *>(&NAME##_u8_db10_win);              \
        e.fn[fsea::K_U8_WIN] = reinterpret_cast<const void *>(&NAME##_u8_win);                        \
        e.name[fsea::K_U8_MAG_WIN] = #NAME "_u8_mag_win";                                             \
        e.name[fsea::K_U8_DB5_WIN] = #NAME "_u8_db5_win";                                             \
        e.name[fsea::K_U8_DB10_WIN] = #NAME "_u8_db10_win";                                           \
        e.name[fsea::K_U8_WIN] = #NAME "_u8_win";                                                     \
        e.launch_win = &NAME##_launch_win;                                                            \
    }
#define FSEA_DEFINE_HALF_OVERLAP_WIN(NAME, WMODE)                                                     \
    FSEA_KERNEL_FN_(NAME, _u8_mag_half_win, fsea::IN_U8, fsea::MODE_MAG, false, true, WMODE)          \
    static void NAME##_launch_win_all(int kind, const fsea::FftArgs &a, unsigned grid, hipStream_t s) { \
        if (kind == fsea::K_U8_MAG_HALF_WIN) hipLaunchKernelGGL(NAME##_u8_mag_half_win, dim3(grid), dim3(NAME##_cfg::WG), 0, s, a); \
        else NAME##_launch_win(kind, a, grid, s);                                                     \
    }                                                                                                 \
    static void NAME##_add_half_win(fsea::KernelEntry &e) {                                           \
        NAME##_add_win(e);                                                                            \
        e.fn[fsea::K_U8_MAG_HALF_WIN] = reinterpret_cast<const void *>(&NAME##_u8_mag_half_win);      \
        e.name[fsea::K_U8_MAG_HALF_WIN] = #NAME "_u8_mag_half_win";                                   \
        e.launch_win = &NAME##_launch_win_all;                                                        \
    }
#define FSEA_REGISTER_WIN(NAME) if (n < cap) { out[n] = NAME##_entry(); NAME##_add_win(out[n]); ++n; }
#define FSEA_REGISTER_HALF_WIN(NAME) if (n < cap) { out[n] = NAME##_entry_half(); NAME##_add_half_win(out[n]); ++n; }

#define FSEA_REGISTER_BEGIN(TAG)                                                                      \
    extern "C" int fsea_kernels_##TAG(fsea::KernelEntry *out, int cap) {                              \
        int n = 0;
#define FSEA_REGISTER(NAME) if (n < cap) out[n++] = NAME##_entry();
#define FSEA_REGISTER_END                                                                             \
        return n;                                                                                     \
    }
