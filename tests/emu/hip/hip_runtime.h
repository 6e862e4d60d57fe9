// tests/emu/hip/hip_runtime.h -- TEST-ONLY stand-in for the HIP device runtime so that
// frequensea_amd/csrc/fsea_fft_core.h (the real kernel source, unmodified) can be compiled
// with g++ and executed on the CPU: one std::thread per work-item, __syncthreads() as a
// std::barrier.  This checks the index arithmetic (Stockham addressing, LDS padding,
// twiddle tables, epilogue ownership of bins) in the no-GPU test tier.  It is not a
// product path and nothing in frequensea_amd/ includes it.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

struct emu_dim3 { unsigned x = 1, y = 1, z = 1; };
extern thread_local emu_dim3 threadIdx, blockIdx, blockDim, gridDim;

void emu_syncthreads();
#define __syncthreads() emu_syncthreads()
#define __builtin_amdgcn_wave_barrier() emu_syncthreads()
#define __builtin_amdgcn_fence(order, scope) ((void)0)
// clang vector builtins used by the packed-math code, for g++ vector_size types
template <class V>
static inline V emu_elementwise_fma(V a, V b, V c) {
    V r;
    for (unsigned i = 0; i < sizeof(V) / sizeof(float); ++i) r[i] = fmaf(a[i], b[i], c[i]);
    return r;
}
#define __builtin_elementwise_fma(a, b, c) emu_elementwise_fma(a, b, c)
template <class V>
static inline V emu_shuffle2(V a, V b, int i, int j) {
    const float src[4] = {a[0], a[1], b[0], b[1]};
    return V{src[i], src[j]};
}
#define __builtin_shufflevector(a, b, i, j) emu_shuffle2(a, b, i, j)
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)
#define __builtin_amdgcn_logf(x) log2f(x)
