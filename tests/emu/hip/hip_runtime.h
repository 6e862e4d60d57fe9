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

void emu_syncthreads();   // all work-items of the emulated workgroup
void emu_wave_barrier();  // the 64 work-items of this work-item's wavefront
unsigned emu_readfirstlane(unsigned v);
#define __syncthreads() emu_syncthreads()
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier()
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane(v)
#define __builtin_amdgcn_s_waitcnt(n) ((void)0)
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) __atomic_load_n(p, order)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n(p, v, order)
static inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
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
// buffer resources: base + byte count, with the hardware's range check (out-of-range loads
// return zero, out-of-range stores are dropped)
#include <cstring>
struct emu_rsrc { char *base; uint32_t n; };
typedef emu_rsrc __amdgpu_buffer_rsrc_t;
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, n, flags) emu_rsrc{(char *)(p), (uint32_t)(n)}
template <class T>
static inline T emu_bload(emu_rsrc rs, uint32_t voff, uint32_t soff) {
    T v;
    std::memset(&v, 0, sizeof(T));
    const uint64_t off = (uint64_t)voff + soff;
    if (off + sizeof(T) <= rs.n) std::memcpy(&v, rs.base + off, sizeof(T));
    return v;
}
template <class T>
static inline void emu_bstore(T v, emu_rsrc rs, uint32_t voff, uint32_t soff) {
    const uint64_t off = (uint64_t)voff + soff;
    if (off + sizeof(T) <= rs.n) std::memcpy(rs.base + off, &v, sizeof(T));
}
typedef uint32_t emu_u32x2 __attribute__((vector_size(8)));
typedef uint32_t emu_u32x4 __attribute__((vector_size(16)));
// variadic so that brace-initialised vector arguments (commas) pass through the preprocessor
static inline uint16_t emu_bl16(emu_rsrc rs, uint32_t v, uint32_t s, int) { return emu_bload<uint16_t>(rs, v, s); }
static inline uint32_t emu_bl32(emu_rsrc rs, uint32_t v, uint32_t s, int) { return emu_bload<uint32_t>(rs, v, s); }
static inline emu_u32x2 emu_bl64(emu_rsrc rs, uint32_t v, uint32_t s, int) { return emu_bload<emu_u32x2>(rs, v, s); }
static inline emu_u32x4 emu_bl128(emu_rsrc rs, uint32_t v, uint32_t s, int) { return emu_bload<emu_u32x4>(rs, v, s); }
static inline void emu_bs8(uint8_t d, emu_rsrc rs, uint32_t v, uint32_t s, int) { emu_bstore<uint8_t>(d, rs, v, s); }
static inline void emu_bs16(uint16_t d, emu_rsrc rs, uint32_t v, uint32_t s, int) { emu_bstore<uint16_t>(d, rs, v, s); }
static inline void emu_bs32(uint32_t d, emu_rsrc rs, uint32_t v, uint32_t s, int) { emu_bstore<uint32_t>(d, rs, v, s); }
static inline void emu_bs64(emu_u32x2 d, emu_rsrc rs, uint32_t v, uint32_t s, int) { emu_bstore<emu_u32x2>(d, rs, v, s); }
static inline void emu_bs128(emu_u32x4 d, emu_rsrc rs, uint32_t v, uint32_t s, int) { emu_bstore<emu_u32x4>(d, rs, v, s); }
#define __builtin_amdgcn_raw_buffer_load_b16(...) emu_bl16(__VA_ARGS__)
#define __builtin_amdgcn_raw_buffer_load_b32(...) emu_bl32(__VA_ARGS__)
#define __builtin_amdgcn_raw_buffer_load_b64(...) emu_bl64(__VA_ARGS__)
#define __builtin_amdgcn_raw_buffer_load_b128(...) emu_bl128(__VA_ARGS__)
#define __builtin_amdgcn_raw_buffer_store_b8(...) emu_bs8(__VA_ARGS__)
#define __builtin_amdgcn_raw_buffer_store_b16(...) emu_bs16(__VA_ARGS__)
#define __builtin_amdgcn_raw_buffer_store_b32(...) emu_bs32(__VA_ARGS__)
#define __builtin_amdgcn_raw_buffer_store_b64(...) emu_bs64(__VA_ARGS__)
#define __builtin_amdgcn_raw_buffer_store_b128(...) emu_bs128(__VA_ARGS__)
static inline unsigned long long wall_clock64() { return 0; }
#define __builtin_readcyclecounter() 0ull
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_s_setprio(n) ((void)0)
#define __builtin_amdgcn_sched_barrier(n) ((void)0)
static inline void sincospi(double x, double *s, double *c) {
    *s = sin(3.14159265358979323846 * x);
    *c = cos(3.14159265358979323846 * x);
}
static inline void sincospif(float x, float *s, float *c) {
    *s = sinf(3.14159265358979323846f * x);
    *c = cosf(3.14159265358979323846f * x);
}
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)
#define __builtin_amdgcn_logf(x) log2f(x)
