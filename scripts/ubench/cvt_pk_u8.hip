// What does v_cvt_pk_u8_f32 do to out-of-range and fractional inputs on gfx950?  The pixel
// epilogues need C's (int) truncation toward zero followed by a clamp to [0, 255]
// (c/fft-batch.c:35-37, 86-90).  Build: hipcc --offload-arch=gfx950 -O2 cvt_pk_u8.hip -o cvt_pk_u8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

__global__ void k(const float *in, unsigned *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0u, 0u);
}

int main() {
    std::vector<float> v = {-INFINITY, -300.f, -1.5f, -1.f, -0.75f, -0.5f, -0.25f, -0.f, 0.f, 0.25f, 0.49999f, 0.5f, 0.50001f, 0.75f,
                            0.99999f, 1.f, 1.5f, 2.5f, 3.5f, 3.99999f, 126.5f, 127.5f, 254.49f, 254.5f, 254.99f, 255.f, 255.4f,
                            255.5f, 255.99f, 256.f, 300.f, 1e9f, INFINITY, NAN};
    float *d_in; unsigned *d_out;
    hipMalloc(&d_in, v.size() * 4); hipMalloc(&d_out, v.size() * 4);
    hipMemcpy(d_in, v.data(), v.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out, (int)v.size());
    std::vector<unsigned> o(v.size());
    hipMemcpy(o.data(), d_out, v.size() * 4, hipMemcpyDeviceToHost);
    int trunc_ok = 1, rne_ok = 1;
    for (size_t i = 0; i < v.size(); ++i) {
        float x = v[i];
        int want_trunc = std::isnan(x) ? 0 : (x <= 0.f ? 0 : (x >= 255.f ? 255 : (int)x));
        float r = nearbyintf(x);
        int want_rne = std::isnan(x) ? 0 : (r <= 0.f ? 0 : (r >= 255.f ? 255 : (int)r));
        printf("%12g -> %3u   (trunc+clamp %3d, rne+clamp %3d)\n", x, o[i] & 0xff, want_trunc, want_rne);
        if ((int)(o[i] & 0xff) != want_trunc) trunc_ok = 0;
        if ((int)(o[i] & 0xff) != want_rne) rne_ok = 0;
    }
    printf("matches truncation+clamp: %s; matches round-to-nearest-even+clamp: %s\n", trunc_ok ? "yes" : "no", rne_ok ? "yes" : "no");
    return 0;
}
