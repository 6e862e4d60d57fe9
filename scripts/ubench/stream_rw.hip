// Microbenchmark: what HBM rate does a plain streaming kernel reach on this chip for the FFT
// path's traffic mix (1 byte read : 2 bytes written) and for a 1:1 copy?  16 B per lane accesses,
// grid-stride, data set far larger than the 256 MiB Infinity Cache.
// Build: hipcc --offload-arch=gfx950 -O3 stream_rw.hip -o stream_rw
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_1r2w(const uint4 *in, float4 *out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = in[i];
        out[2 * i] = make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
        out[2 * i + 1] = make_float4((float)(v.x >> 8), (float)(v.y >> 8), (float)(v.z >> 8), (float)(v.w >> 8));
    }
}
__global__ void k_copy(const uint4 *in, uint4 *out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void k_write(float4 *out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void k_read(const uint4 *in, unsigned *sink, size_t n16) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = in[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const size_t in_bytes = (size_t)1 << 30, out_bytes = (size_t)2 << 30;
    void *d_in, *d_out; unsigned *d_sink;
    (void)hipMalloc(&d_in, in_bytes); (void)hipMalloc(&d_out, out_bytes); (void)hipMalloc(&d_sink, 4);
    (void)hipMemset(d_in, 1, in_bytes); (void)hipMemset(d_out, 0, out_bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int blocks : {2048, 4096, 8192}) {
        for (int kind = 0; kind < 4; ++kind) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                (void)hipEventRecord(e0);
                const size_t n16 = in_bytes / 16;
                if (kind == 0) hipLaunchKernelGGL(k_1r2w, dim3(blocks), dim3(256), 0, 0, (const uint4 *)d_in, (float4 *)d_out, n16);
                if (kind == 1) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, (const uint4 *)d_in, (uint4 *)d_out, n16);
                if (kind == 2) hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, (float4 *)d_out, out_bytes / 16);
                if (kind == 3) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, (const uint4 *)d_in, d_sink, n16);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double bytes = kind == 0 ? 3.0 * in_bytes : kind == 1 ? 2.0 * in_bytes : kind == 2 ? (double)out_bytes : (double)in_bytes;
            const char *name[] = {"1 read : 2 written", "copy 1:1", "write only", "read only"};
            printf("blocks=%-5d %-20s %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)\n", blocks, name[kind], best, bytes / best / 1e6, bytes / best / 1e6 / 80.0);
        }
    }
    return 0;
}
