// Microbenchmark: issue rate of scalar vs packed f32 VALU instructions on gfx950, as a
// function of waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
// Each wave runs ITER iterations of 16 independent instructions of one kind (inline asm so the
// compiler cannot fuse or pack them) and reports s_memtime cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096

#define BODY16(OP)                                                                            \
    OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

template <int KIND>
__global__ void k(float *out, unsigned long long *cycles) {
    float a[16];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[16];
    const float s = 1.0001f + threadIdx.x * 1e-9f;
    const f2 s2 = {s, s};
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = i + threadIdx.x; p[i] = f2{(float)i, (float)threadIdx.x}; }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
        if constexpr (KIND == 0) {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            BODY16(OP)
#undef OP
        } else if constexpr (KIND == 1) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
            BODY16(OP)
#undef OP
        } else if constexpr (KIND == 2) {
#define OP(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
            BODY16(OP)
#undef OP
        } else if constexpr (KIND == 3) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(s2));
            BODY16(OP)
#undef OP
        } else if constexpr (KIND == 4) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
            BODY16(OP)
#undef OP
        } else if constexpr (KIND == 5) {
#define OP(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(s));
            BODY16(OP)
#undef OP
        } else if constexpr (KIND == 6) {
#define OP(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            BODY16(OP)
#undef OP
        } else if constexpr (KIND == 7) {
#define OP(i) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "+v"(a[i]));
            BODY16(OP)
#undef OP
        } else if constexpr (KIND == 8) {  // packed fma with op_sel swap + neg (complex-multiply half)
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "+v"(p[i]) : "v"(s2));
            BODY16(OP)
#undef OP
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name, int waves_per_simd, float *d_out, unsigned long long *d_cyc) {
    const int cus = 256;
    const int block = 256;                       // 4 waves = one per SIMD
    const int grid = cus * waves_per_simd;       // waves_per_simd blocks per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(block), 0, 0, d_out, d_cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(block), 0, 0, d_out, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> cyc(grid);
    hipMemcpy(cyc.data(), d_cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : cyc) mean += (double)c;
    mean /= grid;
    const double insts_per_wave = 16.0 * ITER;
    const double wave_insts_per_simd = insts_per_wave * waves_per_simd;  // all waves of one SIMD
    printf("%-22s waves/SIMD=%d  %.3f ms  cyc/inst/wave=%.2f  SIMD cyc per wave-inst=%.2f  clock~%.2f GHz  Gwaveinst/s=%.1f\n",
           name, waves_per_simd, ms, mean / insts_per_wave, mean / wave_insts_per_simd,
           mean / (ms * 1e-3) / 1e9, (double)grid * 4 * insts_per_wave / (ms * 1e-3) / 1e9);
}

int main() {
    float *d_out;
    unsigned long long *d_cyc;
    hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    hipMalloc(&d_cyc, 256 * 8 * sizeof(unsigned long long));
    for (int w : {1, 2, 4}) {
        run<0>("v_add_f32", w, d_out, d_cyc);
        run<1>("v_fma_f32", w, d_out, d_cyc);
        run<5>("v_fmac_f32", w, d_out, d_cyc);
        run<2>("v_pk_add_f32", w, d_out, d_cyc);
        run<4>("v_pk_mul_f32", w, d_out, d_cyc);
        run<3>("v_pk_fma_f32", w, d_out, d_cyc);
        run<8>("v_pk_fma_f32 opsel/neg", w, d_out, d_cyc);
        run<6>("v_sqrt_f32", w, d_out, d_cyc);
        run<7>("v_cvt_f32_i32_sdwa", w, d_out, d_cyc);
    }
    return 0;
}
