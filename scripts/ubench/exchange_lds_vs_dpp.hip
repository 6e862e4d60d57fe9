// Microbenchmark for the experiment BASELINE.json's north_star names ("wavefront-shuffle twiddle
// exchange"): the intra-wave part of a Stockham pass exchange done with cross-lane VALU operations
// (DPP) instead of an LDS round trip.
//
// The exchange is the one a single-wave frame needs between two radix-16 passes: inside every
// 16-lane row, lane b holds 16 items (16 bytes = two complex each) indexed ka and has to end up with
// item b of every lane ka -- a 16 x 16 transpose of 16-byte items, 64 VGPRs per lane.
//   MODE 0  LDS: 16 ds_write_b128 + 16 ds_read_b128 per lane, wave-private region, no barrier
//           (the conflict-free layout slot = 65 ka + 16 g + b of fsea_fft_core.h run_v2)
//   MODE 1  DPP: four butterfly stages (lane ^ 1, ^ 2, ^ 4, ^ 8); per stage every lane keeps half of
//           its items and swaps the other half with its partner lane
// Each wave runs ITER exchanges back to back (a dependent chain, like the passes of a frame, with
// one packed FMA per item in between so that the compiler cannot collapse consecutive exchanges);
// 8 waves per CU = 2 per SIMD as in the FFT kernels.  Prints ns per exchange per wave and the VALU /
// LDS instruction counts of one exchange (from the ISA, see profiles/r02_exchange_dpp_vs_lds.txt).
// Build: hipcc --offload-arch=gfx950 -O3 exchange_lds_vs_dpp.hip -o bin/exchange_lds_vs_dpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITER 2000
typedef float f4 __attribute__((ext_vector_type(4)));

template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_upd(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                 CTRL, 0xf, BANK_MASK, false));
}

// one butterfly stage of the transpose: items r and r | S swap between lane and lane ^ S
template <int S>
__device__ __forceinline__ void stage(f4 (&x)[16], int lane) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (r & S) continue;
        f4 &a = x[r], &b = x[r | S];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float ta = a[c], tb = b[c];
            if constexpr (S == 8) {          // lane ^ 8 = row_ror:8; lanes 8..15 of a row are banks 2, 3
                a[c] = dpp_upd<0x128, 0xc>(ta, tb);   // lanes with bit 3 set take the partner's b
                b[c] = dpp_upd<0x128, 0x3>(tb, ta);   // lanes with bit 3 clear take the partner's a
            } else if constexpr (S == 4) {   // lane + 4 = row_shl:4 (banks 0, 2), lane - 4 = row_shr:4 (banks 1, 3)
                a[c] = dpp_upd<0x114, 0xa>(ta, tb);
                b[c] = dpp_upd<0x104, 0x5>(tb, ta);
            } else {                         // inside a quad: quad_perm + per-lane select
                constexpr int CTRL = (S == 1) ? 0xB1 : 0x4E;   // [1,0,3,2] / [2,3,0,1]
                const float pa = dpp_upd<CTRL, 0xf>(ta, ta), pb = dpp_upd<CTRL, 0xf>(tb, tb);
                const bool hi = (lane & S) != 0;
                a[c] = hi ? pb : ta;
                b[c] = hi ? tb : pa;
            }
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float *out, unsigned long long *cycles) {
    __shared__ f4 lds[8 * 1088];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = lane & 15, g = lane >> 4;
    f4 x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = f4{(float)(r + 16 * lane), 1.0f, (float)lane, (float)r};
    f4 *wr = lds + 1088 * w + 16 * g + b;        // + 65 ka
    f4 *rd = lds + 1088 * w + 65 * b + 16 * g;   // + b'
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) wr[65 * r] = x[r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = rd[r];
        } else {
            stage<1>(x, lane);
            stage<2>(x, lane);
            stage<4>(x, lane);
            stage<8>(x, lane);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = x[r] * 1.0000001f + 1e-9f;   // the "pass" between two exchanges
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += x[r][0] + x[r][1] + x[r][2] + x[r][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) cycles[blockIdx.x * 8 + w] = t1 - t0;
}

// correctness: after one exchange lane (b, g) must hold item b of lane (ka, g) in slot ka
template <int MODE>
__global__ __launch_bounds__(64) void check(int *bad) {
    __shared__ f4 lds[1088];
    const int lane = threadIdx.x, b = lane & 15, g = lane >> 4;
    f4 x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = f4{(float)(r + 16 * lane), 0.f, 0.f, 0.f};
    if constexpr (MODE == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[65 * r + 16 * g + b] = x[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = lds[65 * b + 16 * g + r];
    } else {
        stage<1>(x, lane);
        stage<2>(x, lane);
        stage<4>(x, lane);
        stage<8>(x, lane);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float want = (float)(b + 16 * (r + 16 * g));   // item b of lane (r, g)
        if (x[r][0] != want) atomicAdd(bad, 1);
    }
}

int main() {
    int dev_cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) dev_cus = prop.multiProcessorCount;
    const int blocks = dev_cus;  // one 512-thread block per CU = 8 waves = 2 per SIMD
    float *d_out; unsigned long long *d_cyc; int *d_bad;
    hipMalloc(&d_out, sizeof(float) * blocks * 512);
    hipMalloc(&d_cyc, sizeof(unsigned long long) * blocks * 8);
    hipMalloc(&d_bad, sizeof(int));
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(d_bad, 0, sizeof(int));
        if (mode == 0) hipLaunchKernelGGL(check<0>, dim3(1), dim3(64), 0, 0, d_bad);
        else hipLaunchKernelGGL(check<1>, dim3(1), dim3(64), 0, 0, d_bad);
        int bad = -1;
        hipMemcpy(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, d_out, d_cyc);
            else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, d_out, d_cyc);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        std::vector<unsigned long long> cyc(blocks * 8);
        hipMemcpy(cyc.data(), d_cyc, cyc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto c : cyc) mean += (double)c;
        mean /= (double)cyc.size();
        printf("%-4s transpose errors %d | kernel %.3f ms for %d exchanges per wave, 8 waves/CU: %.1f ns and %.0f shader cycles per "
               "exchange per wave (incl. 16 packed-FMA-equivalents of filler)\n",
               mode == 0 ? "LDS" : "DPP", bad, best, ITER, 1e6 * best / ITER, mean / ITER);
    }
    return 0;
}
