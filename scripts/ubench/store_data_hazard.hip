// How many wait states does gfx950 need between a 16-byte store and a VALU write to one of its data registers?
// Each wavefront stores four known dwords per lane (global_store_dwordx4 ... nt) and overwrites the FIRST data register K
// VALU instructions later, thousands of times, with every CU fully occupied; the host counts dwords that hold the
// overwriting value instead of the stored one and reports which lanes they sit in.  LLVM's hazard recognizer keeps 2 wait
// states for this (VmemStoreHazard, gfx940 and up); the FFT kernels' corrupted rows (profiles/r02_store_data_hazard.txt)
// say that is not always enough.  Build: hipcc --offload-arch=gfx950 -O2 store_data_hazard.hip -o store_data_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// the FFT kernels' form: buffer_store_dwordx4 with a VGPR offset (offen), an SGPR offset and nt
template <int K, bool HOGS = false, bool SALU = false>
__global__ __launch_bounds__(256) void kb(uint4 *out, int iters, unsigned long long total_bytes) {
    const unsigned lane = threadIdx.x, wave_global = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    if (HOGS && (blockIdx.x & 1)) {
        // the neighbours the FFT kernels have: wavefronts issuing packed FMAs with three 64-bit register operands back to back
        for (int it = 0; it < iters * 24; ++it) {
            asm volatile(".rept 16\n\tv_pk_fma_f32 v[30:31], v[32:33], v[34:35], v[36:37]\n\tv_pk_fma_f32 v[38:39], v[40:41], v[42:43], v[44:45]\n\t.endr"
                         ::: "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45");
        }
        return;
    }
    const unsigned long long base = (unsigned long long)out;
    const u32x4 rsrc = {(unsigned)base, (unsigned)(base >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
    (void)total_bytes;
    for (int it = 0; it < iters; ++it) {
        const unsigned a = 0x10000000u | (it << 8) | (lane & 255u), b = a + 1, c = a + 2, d = a + 3;
        const unsigned voff = (lane & 63) * 16u;
        const unsigned soff = (unsigned)(((size_t)wave_global * iters + it) * 64 * 16);   // uniform per wave: an SGPR
        asm volatile(
            "v_mov_b32 v20, %[a]\n\tv_mov_b32 v21, %[b]\n\tv_mov_b32 v22, %[c]\n\tv_mov_b32 v23, %[d]\n\t"
            "s_nop 7\n\t"
            "buffer_store_dwordx4 v[20:23], %[voff], %[rsrc], %[soff] offen nt\n\t"
            ".if %c[salu]\n\t.rept %c[k]\n\ts_mov_b32 s40, s40\n\t.endr\n\t.else\n\t.rept %c[k]\n\tv_fma_f32 v24, v25, v25, v26\n\t.endr\n\t.endif\n\t"
            "v_fma_f32 v20, v25, v25, v26\n\t"
            "s_nop 7\n\t"
            :
            : [a] "v"(a), [b] "v"(b), [c] "v"(c), [d] "v"(d), [voff] "v"(voff), [rsrc] "s"(rsrc),
              [soff] "s"(__builtin_amdgcn_readfirstlane(soff)), [k] "n"(K), [salu] "n"(SALU ? 1 : 0)
            : "v20", "v21", "v22", "v23", "v24", "s40", "memory");
    }
}

template <int K>
__global__ __launch_bounds__(256) void k(uint4 *out, int iters) {
    const unsigned lane = threadIdx.x, wave_global = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    for (int it = 0; it < iters; ++it) {
        const unsigned a = 0x10000000u | (it << 8) | (lane & 255u), b = a + 1, c = a + 2, d = a + 3;
        uint4 *p = out + ((size_t)wave_global * iters + it) * 64 + (lane & 63);
        asm volatile(
            "v_mov_b32 v20, %[a]\n\tv_mov_b32 v21, %[b]\n\tv_mov_b32 v22, %[c]\n\tv_mov_b32 v23, %[d]\n\t"
            "s_nop 7\n\t"
            "global_store_dwordx4 %[p], v[20:23], off nt\n\t"
            ".rept %c[k]\n\tv_mov_b32 v24, v24\n\t.endr\n\t"      // K independent VALU instructions = K wait states
            "v_mov_b32 v20, 0x7fc00000\n\t"                        // the early writer
            "s_nop 7\n\t"
            :
            : [a] "v"(a), [b] "v"(b), [c] "v"(c), [d] "v"(d), [p] "v"(p), [k] "n"(K)
            : "v20", "v21", "v22", "v23", "v24", "memory");
    }
}

template <int K, bool BUFFER, bool HOGS = false, bool SALU = false>
static void run(uint4 *d_out, std::vector<uint4> &h, int blocks, int iters) {
    hipMemset(d_out, 0, h.size() * sizeof(uint4));
    if (BUFFER) hipLaunchKernelGGL((kb<K, HOGS, SALU>), dim3(blocks), dim3(256), 0, 0, d_out, iters, (unsigned long long)(h.size() * sizeof(uint4)));
    else hipLaunchKernelGGL(k<K>, dim3(blocks), dim3(256), 0, 0, d_out, iters);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_out, h.size() * sizeof(uint4), hipMemcpyDeviceToHost);
    size_t bad = 0, other = 0;
    size_t by_quarter[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < h.size(); ++i) {
        const unsigned lane = (unsigned)(i & 63), it = (unsigned)((i / 64) % iters);
        const unsigned want = 0x10000000u | (it << 8) | (((i / 64 / iters) % 4) * 64 + lane);
        if (HOGS && ((i / 64 / iters / 4) & 1)) continue;            // a hog block: nothing stored
        if (h[i].x != want) { ++bad; ++by_quarter[(lane % 16) / 4]; }
        if (!(HOGS && ((i / 64 / iters / 4) & 1)) && (h[i].y != want + 1 || h[i].z != want + 2 || h[i].w != want + 3)) ++other;
    }
    printf("%s K=%d wait states: %zu of %zu stores wrote something else in dword 0 (lanes 0-3 / 4-7 / 8-11 / 12-15 of a 16-lane row: %zu / %zu / %zu / %zu); other mismatches %zu\n",
           SALU ? "buffer_store_dwordx4, wait states made of SALU instructions:" : HOGS ? "buffer_store_dwordx4, packed-FMA neighbours on the SIMD:" : BUFFER ? "buffer_store_dwordx4 offen+soffset nt, v_fma fillers:" : "global_store_dwordx4 nt, v_mov fillers:          ", K, bad, h.size(), by_quarter[0], by_quarter[1], by_quarter[2], by_quarter[3], other);
}

int main() {
    const int blocks = 256 * 8, iters = 64;       // 8 workgroups of 4 waves per CU
    std::vector<uint4> h((size_t)blocks * 4 * iters * 64);
    uint4 *d_out;
    if (hipMalloc(&d_out, h.size() * sizeof(uint4)) != hipSuccess) return 1;
    for (int rep = 0; rep < 2; ++rep) {
        run<0, false>(d_out, h, blocks, iters);
        run<1, false>(d_out, h, blocks, iters);
        run<2, false>(d_out, h, blocks, iters);
        run<3, false>(d_out, h, blocks, iters);
        run<0, true>(d_out, h, blocks, iters);
        run<1, true>(d_out, h, blocks, iters);
        run<2, true>(d_out, h, blocks, iters);
        run<3, true>(d_out, h, blocks, iters);
        run<4, true>(d_out, h, blocks, iters);
        run<6, true>(d_out, h, blocks, iters);
        run<1, true, true>(d_out, h, blocks, iters);
        run<2, true, true>(d_out, h, blocks, iters);
        run<3, true, true>(d_out, h, blocks, iters);
        run<4, true, true>(d_out, h, blocks, iters);
        run<1, true, false, true>(d_out, h, blocks, iters);
        run<2, true, false, true>(d_out, h, blocks, iters);
        run<3, true, false, true>(d_out, h, blocks, iters);
        run<4, true, false, true>(d_out, h, blocks, iters);
        run<6, true, false, true>(d_out, h, blocks, iters);
        run<8, true, false, true>(d_out, h, blocks, iters);
    }
    hipFree(d_out);
    return 0;
}
