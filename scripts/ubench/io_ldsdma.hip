// io_ldsdma.hip -- the headline kernel's I/O skeleton (8192-point frames: 16 KiB of int8 IQ in, 32 KiB of f32 rows out,
// 256 threads per frame, persistent grid, nt both ways) with the next frame's bytes brought in three ways:
//   0  buffer_load_dword into VGPRs (what the product kernel does: 16 dwords per lane prefetched one frame ahead)
//   1  LDS-DMA, global_load_lds_dword: 16 per lane, straight into a per-workgroup staging buffer, read back with ds_read_b32
//   2  LDS-DMA, global_load_lds_dwordx4: 4 per lane (1 KiB per wave instruction), wave-private mapping, ds_read_b32
// Nothing is transformed: every row is the converted samples (and their pairwise sums), so the three differ only in how the
// bytes reach the registers.  VERDICT r02 item 3(b): is the one gfx950 memory feature the FFT kernels do not use worth
// integrating?  The FFT kernel's own I/O skeleton reaches 74.6 % of 8 TB/s; the product 50-57 %.
// Build: hipcc --offload-arch=gfx950 -O3 -o bin/io_ldsdma io_ldsdma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int N = 8192, T = 256, ROWS_IN = 16, ROWS_OUT = 32;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ float s8f(uint32_t w, int b) { return (float)(int8_t)(uint8_t)(w >> (8 * b)); }

template <int MODE>
__global__ __launch_bounds__(256, 2) void skeleton(const uint8_t *in, float *out, unsigned n_frames) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[2][N / 2 * (MODE ? 1 : 0) + 4];  // 2 x 16 KiB for the DMA modes
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    unsigned u = blockIdx.x;
    uint32_t raw[ROWS_IN];
    auto issue = [&](unsigned f, int buf) {
        if (f >= n_frames) return;
        const uint8_t *src = in + (size_t)f * (2 * N);
        if constexpr (MODE == 0) {
#pragma unroll
            for (int r = 0; r < ROWS_IN; ++r) raw[r] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(src + 4 * t + 1024 * r));
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int r = 0; r < ROWS_IN; ++r) {
                __builtin_amdgcn_global_load_lds((glb_void *)(src + 1024 * r + 256 * w + 4 * l),
                                                 (lds_void *)(&stage[buf][256 * r + 64 * w]), 4, 0, 2);
            }
        } else {
#pragma unroll
            for (int i = 0; i < ROWS_IN / 4; ++i) {  // lane l: row 4i + l/16, 16-byte piece l%16 of this wave's 256-byte slice
                __builtin_amdgcn_global_load_lds((glb_void *)(src + 1024 * (4 * i + (l >> 4)) + 256 * w + 16 * (l & 15)),
                                                 (lds_void *)(&stage[buf][1024 * i + 256 * w]), 16, 0, 2);
            }
        }
    };
    issue(u, 0);
    int buf = 0;
    [[maybe_unused]] bool first = true;
    while (u < n_frames) {
        const unsigned un = u + gridDim.x;
        uint32_t cur[ROWS_IN];
        if constexpr (MODE == 0) {
#pragma unroll
            for (int r = 0; r < ROWS_IN; ++r) cur[r] = raw[r];
            issue(un, 0);
        } else {
            // this wave's DMA of the current frame has landed: everything but the 32 row stores issued behind it has retired
            // (vmcnt counts loads and stores in issue order); the first frame has no stores behind its DMA
            if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            first = false;
            if constexpr (MODE == 1) {
#pragma unroll
                for (int r = 0; r < ROWS_IN; ++r) cur[r] = stage[buf][256 * r + 64 * w + l];
            } else {
#pragma unroll
                for (int r = 0; r < ROWS_IN; ++r) cur[r] = stage[buf][1024 * (r >> 2) + 256 * w + 64 * (r & 3) + l];
            }
            issue(un, buf ^ 1);
            buf ^= 1;
        }
        float *row = out + (size_t)u * N;
#pragma unroll
        for (int r = 0; r < ROWS_IN; ++r) {
            const float a = s8f(cur[r], 0), b = s8f(cur[r], 1), c = s8f(cur[r], 2), d = s8f(cur[r], 3);
            __builtin_nontemporal_store(a * a + b * b, row + t + 256 * (2 * r));
            __builtin_nontemporal_store(c * c + d * d, row + t + 256 * (2 * r + 1));
        }
        u = un;
    }
}

int main() {
    const unsigned frames = 4096;
    const int sets = 6, reps = 120;
    std::vector<uint8_t *> ins(sets);
    std::vector<float *> outs(sets);
    std::vector<uint8_t> host((size_t)frames * 2 * N);
    for (size_t i = 0; i < host.size(); ++i) host[i] = (uint8_t)((i * 2654435761u) >> 13);
    for (int s = 0; s < sets; ++s) {
        CK(hipMalloc(&ins[s], host.size()));
        CK(hipMalloc(&outs[s], (size_t)frames * N * 4));
        CK(hipMemcpy(ins[s], host.data(), host.size(), hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ref((size_t)2 * N), got((size_t)2 * N);
    for (int round = 0; round < 3; ++round) {
        for (int mode = 0; mode < 3; ++mode) {
            auto launch = [&](int s) {
                if (mode == 0) hipLaunchKernelGGL(skeleton<0>, dim3(512), dim3(T), 0, 0, ins[s], outs[s], frames);
                if (mode == 1) hipLaunchKernelGGL(skeleton<1>, dim3(512), dim3(T), 0, 0, ins[s], outs[s], frames);
                if (mode == 2) hipLaunchKernelGGL(skeleton<2>, dim3(512), dim3(T), 0, 0, ins[s], outs[s], frames);
            };
            for (int i = 0; i < 2 * sets; ++i) launch(i % sets);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) launch(i % sets);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            CK(hipMemcpy(got.data(), outs[1] + (size_t)(frames - 2) * N, got.size() * 4, hipMemcpyDeviceToHost));
            if (mode == 0) ref = got;
            bool same = true;
            for (size_t i = 0; i < got.size(); ++i) same = same && got[i] == ref[i];
            const double gb = (double)frames * (2.0 * N + 4.0 * N) / (ms * 1e-3) / 1e9;
            printf("round %d mode %d (%s): %.4f ms per 4096-frame launch, %.1f GB/s = %.1f %% of 8 TB/s, rows %s\n", round, mode,
                   mode == 0 ? "VGPR prefetch, 16 x buffer_load_dword" : (mode == 1 ? "LDS-DMA 16 x dword" : "LDS-DMA 4 x dwordx4"),
                   ms, gb, gb / 80.0, same ? "identical to mode 0" : "DIFFER");
        }
    }
    return 0;
}
