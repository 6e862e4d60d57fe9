// fsea_anysize.hip -- the transform sizes FFTW takes (fftw_plan_dft_1d, src/nrf.c:564) and the gfx950 kernels do not, on
// top of those kernels: Bluestein's algorithm for everything that is not a power of two (and the powers of two below 32),
// a four-step decomposition for powers of two above 16384.  Compatibility paths: a handful of launches per batch through
// work buffers owned by the plan.
#include <cmath>
#include <vector>

#include "fsea_internal.h"
#include "fsea_tables.h"

using namespace fsea_detail;

namespace {

// ---- Bluestein's algorithm: the transform sizes FFTW takes and the power-of-two kernels do not (src/nrf.c:564) ----
// X[k] = conj(w[k]) * sum_j (x[j] conj(w[j])) w[k - j],  w[j] = e^{i pi j^2 / n}: a length-n DFT as a circular convolution
// of length m = 2^p >= 2n - 1, i.e. two forward transforms of the power-of-two kernels (IN_F32 -> COMPLEX_F32) around a
// pointwise product with the precomputed spectrum of the chirp.  The inner kernels apply the reference's (-1)^j centring
// themselves (they are the NUT_BUFFER_F64 branch of nrf_fft_process): in front of the first transform that IS the
// centring of x; in front of the second it is cancelled by a (-1)^k in the pointwise kernel.
// u8 input is transformed as (u - 128) / 256 and the spectrum of the constant 0.5 (1 + i) -- one bin for even n, spread
// over all bins for odd n -- is added from a table computed in double (blu_dc), as the power-of-two kernels restore
// bin n/2 analytically: the large offset never passes through f32 arithmetic.
__global__ void fsea_blu_prep_kernel(const void *in, int in_f32, uint32_t xormask, size_t hop, int n, int m, const fsea::cf *chirp_conj,
                                     fsea::cf *a, size_t n_frames) {
    const size_t total = n_frames * (size_t)m;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / (size_t)m;
        const int j = (int)(i - f * (size_t)m);
        fsea::cf v = fsea::cf{0.0f, 0.0f};
        if (j < n) {
            float re, im;
            if (in_f32) {
                const float *src = static_cast<const float *>(in) + 2 * (f * hop + (size_t)j);
                re = src[0];
                im = src[1];
            } else {
                const uint8_t *src = static_cast<const uint8_t *>(in) + 2 * (f * hop + (size_t)j);
                const uint8_t mask = (uint8_t)xormask;  // 0: raw int8 (flip), 0x80: offset binary
                re = (float)(int8_t)(src[0] ^ mask) * (1.0f / 256.0f);
                im = (float)(int8_t)(src[1] ^ mask) * (1.0f / 256.0f);
            }
            const fsea::cf w = chirp_conj[j];
            v = fsea::cf{re * w[0] - im * w[1], re * w[1] + im * w[0]};
        }
        a[i] = v;
    }
}

// d[k] = (-1)^k conj(A[k] B[k])
__global__ void fsea_blu_mul_kernel(const fsea::cf *A, const fsea::cf *B, fsea::cf *d, int m, size_t n_frames) {
    const size_t total = n_frames * (size_t)m;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % (size_t)m);
        const fsea::cf x = A[i], b = B[k];
        const float re = x[0] * b[0] - x[1] * b[1], im = x[0] * b[1] + x[1] * b[0];
        const float sgn = (k & 1) ? -1.0f : 1.0f;
        d[i] = fsea::cf{sgn * re, -sgn * im};
    }
}

// X[k] = conj(w[k]) conj(E[k]) / m (+ dc[k]), then the plan's epilogue; one thread per output bin
__global__ void fsea_blu_epilogue_kernel(const fsea::cf *E, const fsea::cf *chirp_conj, const fsea::cf *dc, int n, int m, int mode,
                                         void *out, size_t n_frames) {
    const size_t total = n_frames * (size_t)n;
    const float inv_m = 1.0f / (float)m;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / (size_t)n;
        int k = (int)(i - f * (size_t)n);
        const bool patched = (mode == FSEA_MODE_MAG_F32 || mode == FSEA_MODE_DB5_U8_DCFIX);
        if (patched && k == n / 2 && k > 0) k -= 1;  // bin n/2 := bin n/2 - 1 (src/nrf.c:626-628; c/fft-batch-broad.c:115-117)
        const fsea::cf e = E[f * (size_t)m + (size_t)k], w = chirp_conj[k];
        // conj(w_k) conj(e) = conj(w_k e)... with chirp_conj = conj(w): X = chirp_conj[k] * conj(e)
        float re = (w[0] * e[0] + w[1] * e[1]) * inv_m, im = (w[1] * e[0] - w[0] * e[1]) * inv_m;
        if (dc) {
            re += dc[k][0];
            im += dc[k][1];
        }
        const float p = re * re + im * im;
        if (mode == FSEA_MODE_COMPLEX_F32) {
            static_cast<fsea::cf *>(out)[i] = fsea::cf{re, im};
        } else if (mode == FSEA_MODE_DB10_U8 || mode == FSEA_MODE_DB5_U8_DCFIX) {
            const float d = 10.0f * log10f(p + 1.0e-20f) * (mode == FSEA_MODE_DB10_U8 ? 10.0f : 5.0f);
            int q = (int)d;
            q = q < 0 ? 0 : (q > 255 ? 255 : q);
            static_cast<uint8_t *>(out)[i] = (uint8_t)q;
        } else if (mode == FSEA_MODE_DB_F32) {
            static_cast<float *>(out)[i] = 10.0f * log10f(p + 1.0e-20f);
        } else {
            static_cast<float *>(out)[i] = sqrtf(p);
        }
    }
}

// ---- four-step transform: powers of two above 16384 (fftw_plan_dft_1d takes them) on two passes of the kernels ----
// n = N1 N2; sample j = N2 j1 + j2, bin k = k1 + N1 k2:
//   X[k1 + N1 k2] = sum_{j2} W_N2^{j2 k2} ( W_n^{j2 k1} sum_{j1} x[N2 j1 + j2] W_N1^{j1 k1} )
// i.e. N2 transforms of size N1 over the columns, a twiddle, N1 transforms of size N2, and two transposes folded into the
// helper kernels.  Same conventions as the Bluestein path: the inner kernels' own (-1)^j centring is compensated, u8 input
// is transformed as (u - 128) / 256 and the offset-binary DC term (bin n/2: n is even) is added at the end.
__global__ void fsea_fs_prep_kernel(const void *in, int in_f32, uint32_t xormask, size_t hop, int n1, int n2, fsea::cf *a, size_t n_frames) {
    const size_t n = (size_t)n1 * (size_t)n2, total = n_frames * n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / n, r = i - f * n;
        const size_t j2 = r / (size_t)n1, j1 = r - j2 * (size_t)n1;   // a[f][j2][j1]
        const size_t j = (size_t)n2 * j1 + j2;
        float re, im;
        if (in_f32) {
            const float *src = static_cast<const float *>(in) + 2 * (f * hop + j);
            re = src[0];
            im = src[1];
        } else {
            const uint8_t *src = static_cast<const uint8_t *>(in) + 2 * (f * hop + j);
            const uint8_t mask = (uint8_t)xormask;
            re = (float)(int8_t)(src[0] ^ mask) * (1.0f / 256.0f);
            im = (float)(int8_t)(src[1] ^ mask) * (1.0f / 256.0f);
        }
        // the reference's (-1)^j (j and j2 have the same parity: N2 is even) and the inner kernel's own (-1)^{j1}, undone
        const float sgn = ((j2 ^ j1) & 1) ? -1.0f : 1.0f;
        a[i] = fsea::cf{sgn * re, sgn * im};
    }
}

// b[f][k1][j2] = A[f][j2][k1] W_n^{j2 k1} (-1)^{j2}
__global__ void fsea_fs_twiddle_kernel(const fsea::cf *A, const fsea::cf *tw, fsea::cf *b, int n1, int n2, size_t n_frames) {
    const size_t n = (size_t)n1 * (size_t)n2, total = n_frames * n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / n, r = i - f * n;
        const size_t k1 = r / (size_t)n2, j2 = r - k1 * (size_t)n2;   // b[f][k1][j2]
        const fsea::cf x = A[f * n + j2 * (size_t)n1 + k1], w = tw[j2 * (size_t)n1 + k1];
        const float sgn = (j2 & 1) ? -1.0f : 1.0f;
        b[i] = fsea::cf{sgn * (x[0] * w[0] - x[1] * w[1]), sgn * (x[0] * w[1] + x[1] * w[0])};
    }
}

// out[f][k1 + N1 k2] = B[f][k1][k2] (+ DC at bin n/2), then the plan's epilogue
__global__ void fsea_fs_epilogue_kernel(const fsea::cf *B, int add_dc, int n1, int n2, int mode, void *out, size_t n_frames) {
    const size_t n = (size_t)n1 * (size_t)n2, total = n_frames * n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / n;
        size_t k = i - f * n;
        const bool patched = (mode == FSEA_MODE_MAG_F32 || mode == FSEA_MODE_DB5_U8_DCFIX);
        if (patched && k == n / 2) k -= 1;
        const size_t k2 = k / (size_t)n1, k1 = k - k2 * (size_t)n1;
        const fsea::cf e = B[f * n + k1 * (size_t)n2 + k2];
        float re = e[0], im = e[1];
        if (add_dc && k == n / 2) {
            re += 0.5f * (float)n;
            im += 0.5f * (float)n;
        }
        const float p = re * re + im * im;
        if (mode == FSEA_MODE_COMPLEX_F32) {
            static_cast<fsea::cf *>(out)[i] = fsea::cf{re, im};
        } else if (mode == FSEA_MODE_DB10_U8 || mode == FSEA_MODE_DB5_U8_DCFIX) {
            const float d = 10.0f * log10f(p + 1.0e-20f) * (mode == FSEA_MODE_DB10_U8 ? 10.0f : 5.0f);
            int q = (int)d;
            q = q < 0 ? 0 : (q > 255 ? 255 : q);
            static_cast<uint8_t *>(out)[i] = (uint8_t)q;
        } else if (mode == FSEA_MODE_DB_F32) {
            static_cast<float *>(out)[i] = 10.0f * log10f(p + 1.0e-20f);
        } else {
            static_cast<float *>(out)[i] = sqrtf(p);
        }
    }
}

void host_fft(std::vector<double> &re, std::vector<double> &im);

}  // namespace

namespace fsea_detail {

// One Bluestein pass over n_frames frames (in chunks that fit the work buffers): prep -> FFT_m -> x chirp spectrum -> FFT_m ->
// epilogue, all on stream `s`.  d_in: u8 IQ or f32 complex (device, or device-mapped host memory), frame f at sample f * hop.
int blu_launch(fsea_plan *p, int in_kind, const void *d_in, size_t n_frames, int flip, int mode, void *d_out, hipStream_t s) {
    if (n_frames == 0) return FSEA_OK;
    const int n = p->n, m = p->blu_m;
    const size_t esz = mode_elem_bytes(mode);
    const size_t in_bps = (in_kind == fsea::IN_F32) ? 8 : 2;
    for (size_t f0 = 0; f0 < n_frames; f0 += p->blu_work_frames) {
        const size_t nf = (n_frames - f0 < p->blu_work_frames) ? n_frames - f0 : p->blu_work_frames;
        const size_t total = nf * (size_t)m;
        unsigned blocks = (unsigned)((total + 255) / 256);
        if (blocks > 8192) blocks = 8192;
        const char *src = static_cast<const char *>(d_in) + f0 * (size_t)p->hop * in_bps;
        hipLaunchKernelGGL(fsea_blu_prep_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const void *>(src),
                           in_kind == fsea::IN_F32 ? 1 : 0, flip ? 0u : 0x80u, (size_t)p->hop, n, m, p->d_blu_chirp, p->d_blu_work[0], nf);
        int rc = launch(p->blu_inner, fsea::IN_F32, p->d_blu_work[0], nf, 0, FSEA_MODE_COMPLEX_F32, p->d_blu_work[1], s);
        if (rc) return rc;
        hipLaunchKernelGGL(fsea_blu_mul_kernel, dim3(blocks), dim3(256), 0, s, p->d_blu_work[1], p->d_blu_bfft, p->d_blu_work[0], m, nf);
        rc = launch(p->blu_inner, fsea::IN_F32, p->d_blu_work[0], nf, 0, FSEA_MODE_COMPLEX_F32, p->d_blu_work[1], s);
        if (rc) return rc;
        unsigned eblocks = (unsigned)((nf * (size_t)n + 255) / 256);
        if (eblocks > 8192) eblocks = 8192;
        hipLaunchKernelGGL(fsea_blu_epilogue_kernel, dim3(eblocks), dim3(256), 0, s, p->d_blu_work[1], p->d_blu_chirp,
                           in_kind == fsea::IN_F32 ? nullptr : p->d_blu_dc, n, m, mode,
                           static_cast<void *>(static_cast<char *>(d_out) + f0 * (size_t)n * esz), nf);
        FSEA_HIP(hipGetLastError());
    }
    return FSEA_OK;
}

// One four-step pass over n_frames frames (in chunks that fit the work buffers, shared with the Bluestein fields).
int fs_launch(fsea_plan *p, int in_kind, const void *d_in, size_t n_frames, int flip, int mode, void *d_out, hipStream_t s) {
    if (n_frames == 0) return FSEA_OK;
    const int n1 = p->fs_n1, n2 = p->fs_n2;
    const size_t n = (size_t)p->n, esz = mode_elem_bytes(mode);
    const size_t in_bps = (in_kind == fsea::IN_F32) ? 8 : 2;
    for (size_t f0 = 0; f0 < n_frames; f0 += p->blu_work_frames) {
        const size_t nf = (n_frames - f0 < p->blu_work_frames) ? n_frames - f0 : p->blu_work_frames;
        unsigned blocks = (unsigned)((nf * n + 255) / 256);
        if (blocks > 16384) blocks = 16384;
        const char *src = static_cast<const char *>(d_in) + f0 * (size_t)p->hop * in_bps;
        hipLaunchKernelGGL(fsea_fs_prep_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const void *>(src),
                           in_kind == fsea::IN_F32 ? 1 : 0, flip ? 0u : 0x80u, (size_t)p->hop, n1, n2, p->d_blu_work[0], nf);
        int rc = launch(p->fs_inner1, fsea::IN_F32, p->d_blu_work[0], nf * (size_t)n2, 0, FSEA_MODE_COMPLEX_F32, p->d_blu_work[1], s);
        if (rc) return rc;
        hipLaunchKernelGGL(fsea_fs_twiddle_kernel, dim3(blocks), dim3(256), 0, s, p->d_blu_work[1], p->d_fs_tw, p->d_blu_work[0], n1, n2, nf);
        rc = launch(p->fs_inner2, fsea::IN_F32, p->d_blu_work[0], nf * (size_t)n1, 0, FSEA_MODE_COMPLEX_F32, p->d_blu_work[1], s);
        if (rc) return rc;
        hipLaunchKernelGGL(fsea_fs_epilogue_kernel, dim3(blocks), dim3(256), 0, s, p->d_blu_work[1], in_kind == fsea::IN_F32 ? 0 : 1, n1, n2,
                           mode, static_cast<void *>(static_cast<char *>(d_out) + f0 * n * esz), nf);
        FSEA_HIP(hipGetLastError());
    }
    return FSEA_OK;
}

// host-side double FFT (radix 2, in place) for the chirp's spectrum: plan creation only
}  // namespace fsea_detail

namespace {
void host_fft(std::vector<double> &re, std::vector<double> &im) {
    const size_t m = re.size();
    for (size_t i = 1, j = 0; i < m; ++i) {
        size_t bit = m >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            std::swap(re[i], re[j]);
            std::swap(im[i], im[j]);
        }
    }
    for (size_t len = 2; len <= m; len <<= 1) {
        const double ang = -6.283185307179586476925286766559 / (double)len;
        for (size_t i = 0; i < m; i += len) {
            for (size_t k = 0; k < len / 2; ++k) {
                const double wr = std::cos(ang * (double)k), wi = std::sin(ang * (double)k);
                const size_t a = i + k, b = i + k + len / 2;
                const double tr = re[b] * wr - im[b] * wi, ti = re[b] * wi + im[b] * wr;
                re[b] = re[a] - tr;
                im[b] = im[a] - ti;
                re[a] += tr;
                im[a] += ti;
            }
        }
    }
}


}  // namespace

namespace fsea_detail {

// powers of two above 16384 up to FSEA_MAX_FFT_SIZE: n = n1 * n2, both kernel sizes; n1 >= n2
bool fourstep_split(int n, int *n1, int *n2) {
    if (n <= 16384 || n > FSEA_MAX_FFT_SIZE || (n & (n - 1)) != 0) return false;
    int lg = 0;
    while ((1 << lg) < n) ++lg;
    *n1 = 1 << ((lg + 1) / 2);
    *n2 = n / *n1;
    return *n1 <= 16384 && *n2 >= 32;
}

// sizes without a kernel of their own that Bluestein's algorithm covers: m = 2^p >= 2n - 1, itself a kernel size or a
// four-step size (everything that is not a power of two, and the powers of two below 32)
int bluestein_m(int n) {
    if (n < 2 || 2LL * n - 1 > FSEA_MAX_FFT_SIZE) return 0;
    if ((n & (n - 1)) == 0 && n >= 32) return 0;
    int m = 32;
    while (m < 2 * n - 1) m <<= 1;
    return m;
}

// twiddles W_n^{j2 k1} and work buffers of a four-step plan (p->n = fs_n1 * fs_n2)
int fs_setup(fsea_plan *p) {
    const size_t n = (size_t)p->n, n1 = (size_t)p->fs_n1, n2 = (size_t)p->fs_n2;
    std::vector<fsea::TwPair> tw(n);
    const double two_pi = 6.283185307179586476925286766559;
    for (size_t j2 = 0; j2 < n2; ++j2) {
        for (size_t k1 = 0; k1 < n1; ++k1) {
            const double ang = -two_pi * (double)((j2 * k1) % n) / (double)n;   // reduced exactly
            tw[j2 * n1 + k1] = fsea::TwPair{(float)std::cos(ang), (float)std::sin(ang)};
        }
    }
    size_t frames = ((size_t)64 << 20) / (n * sizeof(fsea::cf));
    if (frames < 1) frames = 1;
    p->blu_work_frames = frames;
    FSEA_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_fs_tw), n * sizeof(fsea::cf)));
    FSEA_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_blu_work[0]), frames * n * sizeof(fsea::cf)));
    FSEA_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_blu_work[1]), frames * n * sizeof(fsea::cf)));
    FSEA_HIP(hipMemcpy(p->d_fs_tw, tw.data(), n * sizeof(fsea::cf), hipMemcpyHostToDevice));
    return FSEA_OK;
}

// tables and work buffers of a Bluestein plan (p->n, p->blu_m set, device current)
int blu_setup(fsea_plan *p) {
    const int n = p->n, m = p->blu_m;
    const double pi = 3.14159265358979323846264338327950288;
    std::vector<fsea::TwPair> chirp((size_t)n), dc((size_t)n), bf((size_t)m);
    std::vector<double> br((size_t)m, 0.0), bi((size_t)m, 0.0);
    for (int j = 0; j < n; ++j) {
        const long long q = ((long long)j * (long long)j) % (2LL * n);  // j^2 mod 2n: the phase, reduced exactly
        const double ang = pi * (double)q / (double)n;
        const double wr = std::cos(ang), wi = std::sin(ang);
        chirp[(size_t)j] = fsea::TwPair{(float)wr, (float)-wi};          // conj(w[j])
        br[(size_t)j] = wr;
        bi[(size_t)j] = wi;
        if (j) {
            br[(size_t)(m - j)] = wr;
            bi[(size_t)(m - j)] = wi;
        }
    }
    host_fft(br, bi);
    for (int k = 0; k < m; ++k) bf[(size_t)k] = fsea::TwPair{(float)br[(size_t)k], (float)bi[(size_t)k]};
    // spectrum of the offset-binary DC term 0.5 (1 + i) (-1)^j: n at bin n/2 for even n, 2 / (1 - r_k) for odd n,
    // r_k = e^{i (pi - 2 pi k / n)}
    for (int k = 0; k < n; ++k) {
        double sr = 0.0, si = 0.0;
        if ((n & 1) == 0) {
            if (k == n / 2) sr = (double)n;
        } else {
            const double ang = pi - 2.0 * pi * (double)k / (double)n;
            const double dr = 1.0 - std::cos(ang), di = -std::sin(ang);     // 1 - r
            const double den = dr * dr + di * di;
            sr = 2.0 * dr / den;
            si = -2.0 * di / den;
        }
        dc[(size_t)k] = fsea::TwPair{(float)(0.5 * (sr - si)), (float)(0.5 * (sr + si))};  // 0.5 (1 + i) S
    }
    // work buffers: about 64 MiB each, at least one frame
    size_t frames = ((size_t)64 << 20) / ((size_t)m * sizeof(fsea::cf));
    if (frames < 1) frames = 1;
    if (frames > 65536) frames = 65536;
    p->blu_work_frames = frames;
    FSEA_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_blu_chirp), (size_t)n * sizeof(fsea::cf)));
    FSEA_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_blu_dc), (size_t)n * sizeof(fsea::cf)));
    FSEA_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_blu_bfft), (size_t)m * sizeof(fsea::cf)));
    FSEA_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_blu_work[0]), frames * (size_t)m * sizeof(fsea::cf)));
    FSEA_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_blu_work[1]), frames * (size_t)m * sizeof(fsea::cf)));
    FSEA_HIP(hipMemcpy(p->d_blu_chirp, chirp.data(), (size_t)n * sizeof(fsea::cf), hipMemcpyHostToDevice));
    FSEA_HIP(hipMemcpy(p->d_blu_dc, dc.data(), (size_t)n * sizeof(fsea::cf), hipMemcpyHostToDevice));
    FSEA_HIP(hipMemcpy(p->d_blu_bfft, bf.data(), (size_t)m * sizeof(fsea::cf), hipMemcpyHostToDevice));
    return FSEA_OK;
}


}  // namespace fsea_detail
