// fsea_registry.h -- table of compiled kernel variants (host side).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>

#include "fsea_fft_core.h"

namespace fsea {

struct KernelEntry {
    int n;                 // transform size
    const char *variant;   // "" = default for this n
    int t, fpw, wg, np;
    int radix[4];
    size_t lds_bytes;
    int c0;                // samples per pass-0 load (hop must be a multiple)
    const void *fn_u8_mag; // __global__ function addresses (occupancy queries)
    const void *fn_u8;
    const void *fn_f32;
    const void *fn_u8_rot; // u8 input with the fused frequency shift
    const char *name_u8_mag; // symbol names as rocprof shows them
    const char *name_u8;
    const char *name_f32;
    const char *name_u8_rot;
    void (*launch)(int in_kind, const FftArgs &args, unsigned grid, hipStream_t stream);
};

// Each k_*.hip translation unit exports `int fsea_kernels_<tag>(KernelEntry *out, int cap)`
// which fills `out` with its variants and returns how many it has.

}  // namespace fsea

// Defines the __global__ entry points (u8 IQ: MAG / any mode / frequency-shifted; f32 complex) of one
// configuration, with plain C names so that profiles are easy to read, and the
// launch trampoline + KernelEntry for it.
#define FSEA_DEFINE_KERNEL(NAME, VARIANT, ...)  /* NAME: C symbol stem; VARIANT: registry key */                                                        \
    using NAME##_cfg = fsea::FftCfg<__VA_ARGS__>;                                                     \
    extern "C" __global__ __launch_bounds__(NAME##_cfg::WG, NAME##_cfg::WPE) void NAME##_u8(          \
        fsea::FftArgs a) {                                                                            \
        __shared__ __attribute__((aligned(16))) fsea::cf lds[NAME##_cfg::LDS_ALLOC];                    \
        fsea::FftKernel<NAME##_cfg, fsea::IN_U8>::run(a, lds);                                        \
    }                                                                                                 \
    extern "C" __global__ __launch_bounds__(NAME##_cfg::WG, NAME##_cfg::WPE) void NAME##_u8_mag(      \
        fsea::FftArgs a) {                                                                            \
        __shared__ __attribute__((aligned(16))) fsea::cf lds[NAME##_cfg::LDS_ALLOC];                    \
        fsea::FftKernel<NAME##_cfg, fsea::IN_U8, fsea::MODE_MAG>::run(a, lds);                        \
    }                                                                                                 \
    extern "C" __global__ __launch_bounds__(NAME##_cfg::WG, NAME##_cfg::WPE) void NAME##_f32(         \
        fsea::FftArgs a) {                                                                            \
        __shared__ __attribute__((aligned(16))) fsea::cf lds[NAME##_cfg::LDS_ALLOC];                    \
        fsea::FftKernel<NAME##_cfg, fsea::IN_F32>::run(a, lds);                                       \
    }                                                                                                 \
    extern "C" __global__ __launch_bounds__(NAME##_cfg::WG, NAME##_cfg::WPE) void NAME##_u8_rot(      \
        fsea::FftArgs a) {                                                                            \
        __shared__ __attribute__((aligned(16))) fsea::cf lds[NAME##_cfg::LDS_ALLOC];                    \
        fsea::FftKernel<NAME##_cfg, fsea::IN_U8, -1, true>::run(a, lds);                              \
    }                                                                                                 \
    static void NAME##_launch(int in_kind, const fsea::FftArgs &a, unsigned grid, hipStream_t s) {    \
        if (in_kind == fsea::IN_U8_ROT) {                                                             \
            hipLaunchKernelGGL(NAME##_u8_rot, dim3(grid), dim3(NAME##_cfg::WG), 0, s, a);             \
        } else if (in_kind == fsea::IN_U8 && a.mode == fsea::MODE_MAG && a.xormask == 0) {            \
            hipLaunchKernelGGL(NAME##_u8_mag, dim3(grid), dim3(NAME##_cfg::WG), 0, s, a);             \
        } else if (in_kind == fsea::IN_U8) {                                                          \
            hipLaunchKernelGGL(NAME##_u8, dim3(grid), dim3(NAME##_cfg::WG), 0, s, a);                 \
        } else {                                                                                      \
            hipLaunchKernelGGL(NAME##_f32, dim3(grid), dim3(NAME##_cfg::WG), 0, s, a);                \
        }                                                                                             \
    }                                                                                                 \
    static fsea::KernelEntry NAME##_entry() {                                                         \
        return fsea::KernelEntry{                                                                     \
        NAME##_cfg::N,                                                                                \
        VARIANT,                                                                                      \
        NAME##_cfg::T,                                                                                \
        NAME##_cfg::FPW,                                                                              \
        NAME##_cfg::WG,                                                                               \
        NAME##_cfg::NP,                                                                               \
        {NAME##_cfg::R(0), NAME##_cfg::R(1), NAME##_cfg::R(2), NAME##_cfg::R(3)},                     \
        sizeof(fsea::cf) * NAME##_cfg::LDS_ALLOC,                                                       \
        NAME##_cfg::C(0),                                                                             \
        reinterpret_cast<const void *>(&NAME##_u8_mag),                                               \
        reinterpret_cast<const void *>(&NAME##_u8),                                                   \
        reinterpret_cast<const void *>(&NAME##_f32),                                                  \
        reinterpret_cast<const void *>(&NAME##_u8_rot),                                               \
        #NAME "_u8_mag",                                                                              \
        #NAME "_u8",                                                                                  \
        #NAME "_f32",                                                                                 \
        #NAME "_u8_rot",                                                                              \
        &NAME##_launch};                                                                              \
    }
