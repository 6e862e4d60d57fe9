// fsea_fft_core.h -- device-side building blocks of the gfx950 IQ-FFT kernels.
//
// One workgroup owns FPW frames at a time; a frame is spread over T threads
// with P = N/T points held in registers per thread.  The transform is a
// Stockham autosort FFT in NP passes of radix R[i]: pass 0 reads the 8-bit IQ
// straight from HBM (coalesced, converted in registers, (-1)^n folded into the
// butterflies by operand negation), every pass does an in-register radix-R[i]
// DFT, and passes exchange through one padded LDS buffer.  The last pass fuses
// the magnitude / dB epilogue and writes the output row.
//
// What it replaces in the reference (paths under /root/reference):
//   src/nrf.c:100-109 (byte flip), 599-614 (unpack + (-1)^n), 615 (FFTW),
//   619-630 (magnitude + DC patch); c/fft-batch.c:62-69,83-94;
//   c/fft-batch-broad.c:64-71,106-121.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <fsea_pk_asm.h>  // cf (packed complex), pk_cmul

namespace fsea {

// ---------------------------------------------------------------------------
// compile-time helpers
// ---------------------------------------------------------------------------
constexpr int ilog2c(int n) {
    int l = 0;
    while ((1 << l) < n) ++l;
    return l;
}

constexpr int bitrev_c(int i, int bits) {
    int r = 0;
    for (int b = 0; b < bits; ++b) {
        if (i & (1 << b)) r |= 1 << (bits - 1 - b);
    }
    return r;
}

// LDS index padding (complex units): 2 complexes (16 B) after every P (the
// points one thread holds).  Pass 0 writes one contiguous run of P complexes
// per lane; with the pad consecutive lanes start 16 B further round the banks,
// so the ds_write_b128 of a lane group are conflict-free, and even indices
// stay 16-byte aligned for the vector reads of the later passes.
template <int P>
__host__ __device__ constexpr int lds_pad(int idx) {
    return idx + ((idx / P) << 1);
}

// cos(2 pi m / 64) for the in-register DFT constants.
__host__ __device__ constexpr float cos64q(int i) {
    constexpr float t[17] = {1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f,
                             9.569403357e-01f, 9.238795325e-01f, 8.819212643e-01f,
                             8.314696123e-01f, 7.730104534e-01f, 7.071067812e-01f,
                             6.343932842e-01f, 5.555702330e-01f, 4.713967368e-01f,
                             3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f,
                             9.801714033e-02f, 0.0f};
    return t[i];
}
__host__ __device__ constexpr float cos64(int m) {
    m &= 63;
    if (m <= 16) return cos64q(m);
    if (m <= 32) return -cos64q(32 - m);
    if (m <= 48) return -cos64q(m - 32);
    return cos64q(64 - m);
}
__host__ __device__ constexpr float sin64(int m) { return cos64((m - 16) & 63); }

// ---------------------------------------------------------------------------
// Packed complex arithmetic.  A complex value is one even-aligned VGPR pair
// (cf = 2 x f32) and every butterfly is written so that it selects gfx950's
// packed-f32 VALU ops (v_pk_add/mul/fma_f32: two flops per lane per issue; the
// scalar forms issue at about the same rate, scripts/ubench/valu_rate.hip), with
// swizzles and full-vector negations folded into op_sel / neg modifiers and the
// constant twiddles living in SGPR pairs.
// ---------------------------------------------------------------------------
__device__ __forceinline__ cf cf_swap(cf b) { return __builtin_shufflevector(b, b, 1, 0); }
__device__ __forceinline__ cf cf_fma(cf a, cf b, cf c) { return __builtin_elementwise_fma(a, b, c); }

// radix-2 butterfly with the constant twiddle w = exp(-2 pi i K / L) = (wr, wi):
// (a, b) <- (a + w b, a - w b).  L and K are template arguments: the +-i form is inline assembly,
// and a branch around an asm statement is only removed when its condition is a constant expression.
//   plus  = a + wr * b + (-wi, wi) * swap(b)      2 pk_fma (1 when wr == 0)
//   minus = 2 a - plus                            1 pk_fma
template <int L, int K, bool MI = true>
__device__ __forceinline__ void bfly_const(cf &a, cf &b) {
    if constexpr (K == 0) {
        const cf t = b;
        b = a - t;
        a = a + t;
    } else if constexpr (4 * K == L) {  // w = -i : w b = (b.y, -b.x)
        if constexpr (MI) {
            const cf t = b;
            b = pk_add_pi(a, t);
            a = pk_add_mi(a, t);
        } else {  // the round-1 form: two packed FMAs with (+-1, -+1)
            const cf sb = cf_swap(b);
            b = cf_fma(sb, cf{-1.0f, 1.0f}, a);
            a = cf_fma(sb, cf{1.0f, -1.0f}, a);
        }
    } else {
        constexpr int m = K * (64 / L);
        constexpr float wr = cos64(m), wi = -sin64(m);
        cf plus = cf_fma(b, cf{wr, wr}, a);
        plus = cf_fma(cf_swap(b), cf{-wi, wi}, plus);
        b = cf_fma(a, cf{2.0f, 2.0f}, -plus);
        a = plus;
    }
}

// one level of the constant-twiddle DFT: butterflies (base + k, base + k + HALF), k < HALF
template <int R, int HALF, bool MI, int BASE = 0, int K = 0>
__device__ __forceinline__ void dft_const_level(cf *y) {
    if constexpr (BASE < R) {
        bfly_const<2 * HALF, K, MI>(y[BASE + K], y[BASE + K + HALF]);
        if constexpr (K + 1 < HALF) dft_const_level<R, HALF, MI, BASE, K + 1>(y);
        else dft_const_level<R, HALF, MI, BASE + 2 * HALF, 0>(y);
    }
}
template <int R, int HALF, bool MI>
__device__ __forceinline__ void dft_const_levels(cf *y) {
    if constexpr (HALF < R) {
        dft_const_level<R, HALF, MI>(y);
        dft_const_levels<R, 2 * HALF, MI>(y);
    }
}

// R-point DFT (forward, -1 exponent) over x[0], x[S], ..., x[(R-1)S];
// natural order in and out.  Bit reversal is register renaming only.
template <int R, int S, bool SKIP = false, bool MI = true>
__device__ __forceinline__ void dft_regs(cf *x) {
    if constexpr (SKIP) return;
    static_assert(R >= 2 && R <= 64 && (R & (R - 1)) == 0, "radix must be 2..64");
    constexpr int BITS = ilog2c(R);
    cf y[R];
#pragma unroll
    for (int i = 0; i < R; ++i) y[bitrev_c(i, BITS)] = x[i * S];
    dft_const_levels<R, 1, MI>(y);
#pragma unroll
    for (int i = 0; i < R; ++i) x[i * S] = y[i];
}

// Pass 0 of a windowed kernel: the same DFT over x[0], x[S], ... of samples that still lack their taper weight; weight
// e = E0 + i S of the lane (wv[e / 2], half e % 2; (-1)^n folded in) belongs to x[i S].  The first butterfly level takes the
// weights along: (wa a) +- (wb b) = one packed multiply and two packed FMAs per pair, where weighting the samples first
// costs two multiplies and two adds -- 3 packed ops per pair instead of 4 (2 without a window).  With unit weights the
// results are the plain butterfly's bits: fma(a, 1, b) = a + b.
template <int R, int S, int E0, int I = 0>
__device__ __forceinline__ void dft_win_level(const cf *x, const cf *wv, cf *y) {
    if constexpr (I < R / 2) {  // (template recursion: the weight's half is selected by `if constexpr`, no dead asm left behind)
        constexpr int BITS = ilog2c(R), EA = E0 + I * S, EB = E0 + (I + R / 2) * S;
        const cf a = x[I * S], b = x[(I + R / 2) * S];
        cf t;
        if constexpr (EB & 1) t = pk_scale_hi(b, wv[EB / 2]);
        else t = pk_scale_lo(b, wv[EB / 2]);
        if constexpr (EA & 1) {
            y[bitrev_c(I, BITS)] = pk_wfma_hi(a, wv[EA / 2], t);
            y[bitrev_c(I + R / 2, BITS)] = pk_wfms_hi(a, wv[EA / 2], t);
        } else {
            y[bitrev_c(I, BITS)] = pk_wfma_lo(a, wv[EA / 2], t);
            y[bitrev_c(I + R / 2, BITS)] = pk_wfms_lo(a, wv[EA / 2], t);
        }
        dft_win_level<R, S, E0, I + 1>(x, wv, y);
    }
}
template <int R, int S, int E0, bool MI = true>
__device__ __forceinline__ void dft_regs_win(cf *x, const cf *wv) {
    static_assert(R >= 2 && R <= 64 && (R & (R - 1)) == 0, "radix must be 2..64");
    cf y[R];
    dft_win_level<R, S, E0>(x, wv, y);
    dft_const_levels<R, 2, MI>(y);
#pragma unroll
    for (int i = 0; i < R; ++i) x[i * S] = y[i];
}

// The same DFT with the inter-pass twiddle multiply x[i] *= tw[(i - 1) TS] (i >= 1) fused into
// the first butterfly level, whose own twiddle is 1: with a = x[i] tw_i and b = x[i + R/2] tw_j,
//   plus = a + b = pk_cmul_add(x[i + R/2], tw_j, a)   2 pk_fma
//   minus = 2 a - plus                                  1 pk_fma
// i.e. 5 packed ops per pair instead of 6 (3 instead of 4 for the untwiddled row 0).
// SC0: scale applied to row 0 (the other rows carry it in their twiddles), 1 = none.
template <int R, int S, int TS, bool MI = true>
__device__ __forceinline__ void dft_regs_tw(cf *x, const cf *tw, float sc0) {
    static_assert(R >= 4 && R <= 64 && (R & (R - 1)) == 0, "radix must be 4..64");
    constexpr int BITS = ilog2c(R);
    cf y[R];
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
        cf a = x[i * S];
        if (i == 0) {
            if (sc0 != 1.0f) a = a * cf{sc0, sc0};
        } else {
            a = pk_cmul(a, tw[(i - 1) * TS]);
        }
        const cf plus = pk_cmul_add(x[(i + R / 2) * S], tw[(i + R / 2 - 1) * TS], a);
        y[bitrev_c(i, BITS)] = plus;
        y[bitrev_c(i + R / 2, BITS)] = cf_fma(a, cf{2.0f, 2.0f}, -plus);
    }
    dft_const_levels<R, 2, MI>(y);
#pragma unroll
    for (int i = 0; i < R; ++i) x[i * S] = y[i];
}

// ---- the last level in power form (OPT 8388608; fsea_pk_asm.h) ----
// Where a row only ever leaves as |X|^2 (MAG rows, dB pixels), the last butterfly level of the last pass does not form
// its two complex outputs: for the pair (y[k], y[k + R/2]) with the constant twiddle W_R^k it forms the planar pairs
// (x0.re, x1.re) and (x0.im, x1.im) -- two packed ops for w = 1 / -i, four for a general w, against 2 / 3 -- and from
// them both powers in two packed ops instead of v_mul + v_fmac per bin: one VALU op less per pair with a general
// twiddle, two with a trivial one.  pp[k S] = (|X_k|^2, |X_{k + R/2}|^2).  x0 is bit-identical to bfly_const's;
// x1 = a - w b is formed directly (two roundings) instead of 2 a - x0 (three).
// dc_hi (DC = true, pair 0 only): added to both components of x1 -- the restored offset-binary DC term of bin N/2
// (FftKernel::epilogue), for the one lane and mode that keep it; zero elsewhere.
template <int L, int K, bool DC = false>
__device__ __forceinline__ cf bfly_power(cf a, cf b, float dc_hi = 0.0f) {
    cf re, im;
    if constexpr (K == 0) {
        re = pk_pm_re(a, b);
        im = pk_pm_im(a, b);
    } else if constexpr (4 * K == L) {
        re = pk_pm_re_mi(a, b);
        im = pk_pm_im_mi(a, b);
    } else {
        constexpr int m = K * (64 / L);
        constexpr float wr = cos64(m), wi = -sin64(m);
        const cf w = cf{wr, wi};
        re = pk_pm_re_w(a, b, w);
        im = pk_pm_im_w(a, b, w);
    }
    if constexpr (DC) {
        re[1] += dc_hi;
        im[1] += dc_hi;
    }
    return cf_fma(re, re, im * im);  // the order of the scalar form: fma(re, re, im * im)
}
template <int R, int S, bool DC, int K = 0>
__device__ __forceinline__ void dft_power_level(const cf *y, cf *pp, float dc_hi) {
    if constexpr (K < R / 2) {
        if constexpr (K == 0) pp[0] = bfly_power<R, 0, DC>(y[0], y[R / 2], dc_hi);
        else pp[K * S] = bfly_power<R, K>(y[K], y[K + R / 2]);
        dft_power_level<R, S, DC, K + 1>(y, pp, dc_hi);
    }
}
template <int R, int HALF, int STOP, bool MI>
__device__ __forceinline__ void dft_const_levels_below(cf *y) {
    if constexpr (HALF < STOP) {
        dft_const_level<R, HALF, MI>(y);
        dft_const_levels_below<R, 2 * HALF, STOP, MI>(y);
    }
}
// dft_regs_tw with the last level in power form: x is consumed, pp[k S] (k < R/2) receives the powers
template <int R, int S, int TS, bool MI = true, bool DC = false>
__device__ __forceinline__ void dft_regs_tw_pw(const cf *x, const cf *tw, float sc0, cf *pp, float dc_hi = 0.0f) {
    static_assert(R >= 4 && R <= 64 && (R & (R - 1)) == 0, "radix must be 4..64");
    constexpr int BITS = ilog2c(R);
    cf y[R];
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
        cf a = x[i * S];
        if (i == 0) {
            if (sc0 != 1.0f) a = a * cf{sc0, sc0};
        } else {
            a = pk_cmul(a, tw[(i - 1) * TS]);
        }
        const cf plus = pk_cmul_add(x[(i + R / 2) * S], tw[(i + R / 2 - 1) * TS], a);
        y[bitrev_c(i, BITS)] = plus;
        y[bitrev_c(i + R / 2, BITS)] = cf_fma(a, cf{2.0f, 2.0f}, -plus);
    }
    dft_const_levels_below<R, 2, R / 2, MI>(y);
    dft_power_level<R, S, DC>(y, pp, dc_hi);
}

// The same twiddled DFT with the inter-pass twiddles deferred into the butterfly levels
// (polynomial form): with x_i scaled by om^i (om = W^k, one value per lane), y_k = P(om W_R^k),
// P(z) = sum x_i z^i = E(z^2) + z O(z^2).  The level that merges sub-transforms of size `half`
// then uses the twiddles om^{R/(2 half)} W_{2 half}^j, j < half, of which the upper half is -i
// times the lower one: R/2 run-time twiddles per lane instead of R-1, every butterfly the
// general 3-op form (plus = a + w b in two packed FMAs, minus = 2a - plus in one) -- the same
// 3 R log2 R / 2 packed ops as dft_regs_tw.
// tw layout: [om^{R/2}] [om^{R/4}] [om^{R/8} W_8^{0,1}] [om^{R/16} W_16^{0..3}] ... (fsea_tables.h).
template <int R, int HALF, int TS>
__device__ __forceinline__ void dft_def_level(cf *y, const cf *tw) {
    if constexpr (HALF < R) {
        constexpr int OFF = HALF / 2, DISTINCT = HALF >= 2 ? HALF / 2 : 1;
#pragma unroll
        for (int base = 0; base < R; base += 2 * HALF) {
#pragma unroll
            for (int k = 0; k < DISTINCT; ++k) {
                cf &a = y[base + k], &b = y[base + k + HALF];
                const cf plus = pk_cmul_add(b, tw[(OFF + k) * TS], a);
                b = cf_fma(a, cf{2.0f, 2.0f}, -plus);
                a = plus;
            }
            if constexpr (HALF >= 2) {
#pragma unroll
                for (int k = 0; k < DISTINCT; ++k) {
                    cf &a = y[base + DISTINCT + k], &b = y[base + DISTINCT + k + HALF];
                    const cf plus = pk_cmul_add_mi(b, tw[(OFF + k) * TS], a);
                    b = cf_fma(a, cf{2.0f, 2.0f}, -plus);
                    a = plus;
                }
            }
        }
        dft_def_level<R, 2 * HALF, TS>(y, tw);
    }
}

template <int R, int S, int TS>
__device__ __forceinline__ void dft_regs_def(cf *x, const cf *tw) {
    static_assert(R >= 4 && R <= 64 && (R & (R - 1)) == 0, "radix must be 4..64");
    constexpr int BITS = ilog2c(R);
    cf y[R];
#pragma unroll
    for (int i = 0; i < R; ++i) y[bitrev_c(i, BITS)] = x[i * S];
    dft_def_level<R, 1, TS>(y, tw);
#pragma unroll
    for (int i = 0; i < R; ++i) x[i * S] = y[i];
}

// ---------------------------------------------------------------------------
// configuration
// ---------------------------------------------------------------------------
enum : int {
    MODE_MAG = 0,
    MODE_DB10_U8 = 1,
    MODE_DB5_U8_DCFIX = 2,
    MODE_COMPLEX = 3,
    MODE_MAG_NODC = 4,
    MODE_DB_F32 = 5
};
enum : int { IN_U8 = 0, IN_F32 = 1, IN_U8_ROT = 2 };  // IN_U8_ROT: launch selector only (kernel IN = IN_U8, ROT)

#ifndef FSEA_DEFAULT_OPT
#define FSEA_DEFAULT_OPT 0
#endif
// Per-workgroup time stamps (FftArgs::trace) are compiled into the tuning library only
// (libfsea_hip_tune.so, -DFSEA_TRACE=1); the product kernels carry no trace code.
#ifndef FSEA_TRACE
#define FSEA_TRACE 0
#endif

// N: transform size; T: threads per frame; FPW: frames per workgroup;
// NP passes of radix R0..R3 (unused = 1).  TWL: middle-pass twiddle tables
// live in LDS (else they are fetched from the global table every frame);
// TWR: the last pass keeps its twiddles in registers across the frame loop.
template <int N_, int T_, int FPW_, int WPE_, int NP_, int R0_, int R1_, int R2_ = 1, int R3_ = 1,
          bool TWL_ = true, bool TWR_ = true, int ABL_ = 0, int OPT_ = FSEA_DEFAULT_OPT>
struct FftCfg {
    // OPT: schedule options (all give identical results):
    // 1 = the "everyone has read" barrier of an exchange sits right before the next writes into
    //     the buffer (after the butterflies) instead of right after the reads;
    // 2 = the reads of an exchange stay one batch (scheduling fence behind them), the waits
    //     for them become progressive;
    // 4 = a middle pass fetches all its twiddles from LDS together with the data;
    // 8 = the twiddle multiply is fused into the first butterfly level (dft_regs_tw; one packed
    //     op less per pair, rounding differs in the last bit);
    // 16 = a middle pass that reads 16 bytes per lane (C = 2, P = 32) rotates its lanes inside every
    //     16-lane block by the block index: the pad shifts each block by one 16-byte slot, which
    //     puts two lanes of every ds_read_b128 lane group on one slot (8 instead of 4 LDS cycles
    //     per instruction; scripts/lds_conflicts.py); rotated, the groups are conflict-free;
    // 32 = the same renumbering for the last pass (and the cross-block form for 8-byte layouts, see
    //     pass_lane); on where the last pass has that shape (4096 points), worth nothing in time.
    // 128 = the middle pass's twiddles are deferred into the butterflies (dft_regs_def) and kept in
    //     registers for the workgroup's lifetime: R/2 pairs per column instead of R-1 LDS reads per frame;
    // 256 = the +-i butterflies as packed FMAs by (+-1, -+1) instead of packed adds (the round-1 form);
    // 512 = ticket sizes: the ticket word is waited for behind pass 1's LDS reads, not in front of them;
    // 64 = the V2 schedule (run_v2): first exchange inside each wavefront, two barriers per frame,
    //     middle-pass twiddles deferred into the butterflies and kept in registers.
    // Cache policy: 4096 = row stores nt (where a store writes a whole 128-byte line per frame, st_aux),
    //     8192 = sc1, 16384 = sc0 (tuning), 32768 = input loads nt.
    // Product configurations (fsea_configs.h) use the bits above only.  Tuning-library experiments, all measured
    // and rejected: 1024 = constant higher priority for the younger workgroup of a CU,
    //     2048 = static unit interleave compiled in (the product chooses per launch: FftArgs::dynamic_units),
    //     65536 = the two workgroups of a CU alternate priority per frame, 131072 = priority by the pool's average
    //     progress, 262144 = priority by the partner workgroup's published progress.
    static constexpr int OPT = OPT_;
    // ABL: measurement-only ablations (tuning variants, results are wrong by design):
    // 1 = no output stores, 2 = no LDS exchange / barriers, 4 = no butterflies / twiddles, 8 / 16 / 32 = V2-schedule
    // ablations, 64 = no per-frame loads, 128 = no magnitude arithmetic / no logarithm, 256 = rows stored 16 bytes per
    // lane (misplaced), 512 = frame loaded 16 bytes per lane (misplaced).  Always 0 in the product configurations.
    static constexpr int ABL = ABL_;
    static constexpr int N = N_, T = T_, FPW = FPW_, NP = NP_;
    static constexpr int WPE = WPE_;  // waves per SIMD the register budget must allow
    static constexpr int P = N_ / T_;
    static constexpr int WG = T_ * FPW_;
    static constexpr bool TWL = TWL_, TWR = TWR_;
    static constexpr int R(int i) { return i == 0 ? R0_ : i == 1 ? R1_ : i == 2 ? R2_ : R3_; }
    static constexpr int C(int i) { return P / R(i); }
    static constexpr int Ns(int i) { return i == 0 ? 1 : Ns(i - 1) * R(i - 1); }
    static constexpr int tw_len(int i) { return i == 0 ? 0 : (R(i) - 1) * Ns(i); }
    // LDS (in complex = 8-byte units): FPW padded frames, then the middle-pass tables.
    static constexpr int pad(int idx) { return lds_pad<N_ / T_>(idx); }
    static constexpr int LDS_FRAME = N_ + 2 * T_;
    static constexpr int lds_tw_off(int i) {
        return i <= 1 ? FPW_ * LDS_FRAME : lds_tw_off(i - 1) + tw_len(i - 1);
    }
    // small twiddle block kept in LDS: middle-pass tables, then the two factor tables
    // HI[N/64] = W^{64 h}, LO[64] = W^{l} the last pass's register twiddles are built from
    static constexpr int TAB_MID = lds_tw_off(NP_ - 1) - FPW_ * LDS_FRAME;
    static constexpr int TAB_HI = N_ >= 64 ? N_ / 64 : 1, TAB_LO = 64;
    static constexpr int TAB_SMALL = TAB_MID + (TWR_ ? TAB_HI + TAB_LO : 0);
    static constexpr int LDS_HI = FPW_ * LDS_FRAME + TAB_MID;
    static constexpr int LDS_LO = LDS_HI + TAB_HI;
    static constexpr int LDS_TOTAL = FPW_ * LDS_FRAME + ((TWL_ || TWR_) ? TAB_SMALL : 0);
    static constexpr int LDS_ALLOC = LDS_TOTAL + 2;  // + two ticket words (dynamic frame distribution)
    static_assert(R0_ * R1_ * R2_ * R3_ == N_, "radices must multiply to N");
    static_assert(N_ % T_ == 0, "T must divide N");
    static_assert(NP_ >= 2 && NP_ <= 4, "2..4 passes");
};

struct FftArgs {
    const void *in;       // u8 interleaved IQ, or f32 interleaved complex
    void *out;            // n_frames rows
    size_t n_frames;
    size_t hop;           // samples between frame starts
    uint32_t xormask;     // u8 input: 0 when flip (raw int8), 0x80808080 otherwise
    int mode;             // MODE_*
    unsigned *ctr;        // ticket-counter slot of this launch: [32 q] pool q, [32 * 8] finished workgroups
    unsigned long long *trace;  // diagnostics: [grid][32] = wall start/end, shader-clock start/end, HW_ID, XCC_ID, -, -, end of iteration 0..23; or null
    const cf *tw[4];      // tw[i]: pass-i table, (R_i-1)*Ns_i entries, [r-1][k]
    const cf *tw_small;   // [middle-pass tables | HI | LO], the block copied to LDS (fsea_tables.h)
    const cf *tw_def;     // V2 schedule: deferred middle-pass twiddles, [R0][R1/2] (build_deferred_table)
    // frequency-shifted input (ROT kernels only; fsea_exec_u8_shifted_*): stream sample m is
    // multiplied by e^{2 pi i (rot_phase0 + m rot_delta)}, both in turns.  rot_row[r] =
    // e^{2 pi i rot_delta r N / R0}, the factor between the pass-0 rows of one lane.
    double rot_delta, rot_phase0;
    cf rot_row[32];
    // Tiled output (fsea_exec_u8_tiled_device): the launch's frames are `tile_rows`-row tiles of an image;
    // frame f is written at element (f % tile_rows) * pitch_row + (f / tile_rows) * pitch_tile of `out`
    // instead of f * N, and `out_span` elements are addressable from `out`.  tile_rows == 0: contiguous rows.
    // (V1 schedule only; tile_rows is a multiple of FPW, so a unit never straddles tiles.)
    uint32_t tile_rows = 0, pitch_row = 0, pitch_tile = 0;
    size_t out_span = 0;
    // multi-wave sizes: 1 = units handed out by the ticket pools, 0 = static interleave (unit = blockIdx + k * grid)
    uint32_t dynamic_units = 1;
    // half-overlap kernels (FftKernel<..., RUNS = true>, hop == N/2): a unit is a run of `run_len` consecutive frames, and
    // inside a run the second half of a frame's bytes stays in registers as the first half of the next frame's
    uint32_t run_len = 0;
    // Taper window (WIN kernels; fsea_plan_set_window): x[n] = (-1)^n w[n] u8[n] / 256 -- the weight slot of
    // src/nrf.c:611-612, where the reference has the (-1)^n alone.
    // win: N floats, (-1)^n w[n] in the order the pass-0 lanes hold their samples: weight i = r C0 + c of lane t belongs
    //      to sample n = C0 t + c + r N/R0 and sits at win[(i / 4) 4 T + 4 t + i % 4] (16-byte loads, the lanes' pieces
    //      of one load adjacent).
    // win_dc: the kernels transform w (u8 - 128) and put the offset-binary DC term back as its known spectrum
    //      0.5 (1 + i) D[k], D = DFT of (-1)^n w: 2 NsL entries for the bins [N/2 - NsL, N/2 + NsL), where a cosine-sum taper
    //      has all of it.  A window whose D is not confined to that band (win_offset != 0) is applied to the
    //      offset-binary value itself, w u8, and win_dc is all zero.
    const float *win = nullptr;
    const cf *win_dc = nullptr;
    uint32_t win_offset = 0;
};

// ---------------------------------------------------------------------------
// small typed memory helpers
// ---------------------------------------------------------------------------
template <int C>
__device__ __forceinline__ void ld_c(const cf *p, cf *dst) {
    if constexpr (C == 1) {
        dst[0] = *p;
    } else {
#pragma unroll
        for (int c = 0; c < C; c += 2) {
            const cf2 q = *reinterpret_cast<const cf2 *>(p + c);
            dst[c] = cf{q[0], q[1]};
            dst[c + 1] = cf{q[2], q[3]};
        }
    }
}

template <int C>
__device__ __forceinline__ void st_c(cf *p, const cf *src) {
    if constexpr (C == 1) {
        *p = src[0];
    } else {
#pragma unroll
        for (int c = 0; c < C; c += 2) {
            *reinterpret_cast<cf2 *>(p + c) = cf2{src[c][0], src[c][1], src[c + 1][0], src[c + 1][1]};
        }
    }
}

// ---------------------------------------------------------------------------
// Global memory goes through buffer resources: the 64-bit base and the bounds live
// in four SGPRs, a lane contributes one 32-bit byte offset, and the per-row part of
// the address is a scalar offset -- no 64-bit VALU address arithmetic, and accesses
// outside [0, num_records) return zero / are dropped in hardware, which is what a
// workgroup slot without a frame (ragged last unit) relies on.
// ---------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t buffer_window(const void *base, size_t off, size_t total) {
    const size_t rem = total > off ? total - off : 0;
    char *p = const_cast<char *>(static_cast<const char *>(base)) + (rem ? off : 0);
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, rem > 0xffffffffull ? 0xffffffffu : (uint32_t)rem, 0x00020000);
}

typedef uint32_t u32x2 __attribute__((vector_size(8)));
typedef uint32_t u32x4 __attribute__((vector_size(16)));

__device__ __forceinline__ uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float u2f(uint32_t x) { return __builtin_bit_cast(float, x); }

// 16-byte stores and the registers they read.  A VALU instruction that overwrites a data register of a
// buffer_store_dwordx4 too soon behind it changes what the store writes: the store reads its four registers over several
// cycles, a quarter of each 16-lane row at a time, and lanes 12-15 of every row come last.  LLVM knows the hazard
// (VmemStoreHazard, two wait states on gfx940-class targets) but exempts buffer stores whose soffset is a register -- which
// is where these kernels keep the per-row part of the address -- and on the MI355X the exemption does not hold
// (scripts/ubench/store_data_hazard.hip: with an SGPR soffset and no wait state 1.3 % of the stores are wrong, with two none).
// Unguarded, rows of the 64-, 128- and 2048-point kernels (four adjacent f32 bins per lane) differed between identical
// launches: lanes 12-15 of a row held Im(X)^2 of the NEXT row's bin, the v_mul_f32 that begins the next row's re^2 + im^2
// (scripts/soak.py found it; profiles/r02_store_data_hazard.txt).  Sixteen wait states are spent behind every 16-byte
// store, fenced so that the scheduler cannot move the next writer in front of them.
// FSEA_STORE_GUARD: 1 = that (default), 0 = nothing (the pre-fix code, for the regression evidence: such a build fails
// tests/test_shipped_artifacts.py without a GPU and tests/test_gpu_soak.py / test_gpu_parity.py on one), 2 = two 8-byte
// stores instead (no hazard by construction; 3-20 % slower at those sizes).
#ifndef FSEA_STORE_GUARD
#define FSEA_STORE_GUARD 1
#endif
__device__ __forceinline__ void store_data_guard() {
#if !defined(__AMDGCN__)
    // (the CPU shim of tests/emu compiles this header too: nothing to guard there)
#elif FSEA_STORE_GUARD == 1
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7\n\ts_nop 7");
    __builtin_amdgcn_sched_barrier(0);
#elif FSEA_STORE_GUARD >= 10 /* experiment: FSEA_STORE_GUARD - 10 = operand of a single s_nop */
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop %0" ::"n"(FSEA_STORE_GUARD - 10));
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// THE 16-byte buffer store of these kernels: store + its wait states, one primitive.  Nothing else in the kernel source may
// emit a buffer_store_dwordx3/x4 -- the raw builtin is poisoned right below, so a new store form that bypasses the guard
// does not compile (hipcc; the CPU shim of tests/emu has no hazard and no poison), and tests/test_shipped_artifacts.py
// checks the disassembly of every shipped code object for the wait states behind every 12- and 16-byte store.
template <int AUX>
__device__ __forceinline__ void bst128(rsrc_t rs, uint32_t voff, uint32_t soff, u32x4 data) {
#if FSEA_STORE_GUARD == 2
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{data[0], data[1]}, rs, voff, soff, AUX);
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{data[2], data[3]}, rs, voff + 8, soff, AUX);
#else
    __builtin_amdgcn_raw_buffer_store_b128(data, rs, voff, soff, AUX);
    store_data_guard();
#endif
}
#if defined(__HIP__)
#pragma GCC poison __builtin_amdgcn_raw_buffer_store_b128 __builtin_amdgcn_raw_buffer_store_b96
#endif

// C consecutive f32 / u8 / complex outputs at byte offset voff (+ scalar soff)
// AUX: cache-policy bits of the buffer instruction (0 = default, 2 = nt: streaming, do not keep)
template <int C, int AUX = 0>
__device__ __forceinline__ void bst(rsrc_t rs, uint32_t voff, uint32_t soff, const float *v) {
    if constexpr (C == 1) {
        __builtin_amdgcn_raw_buffer_store_b32(f2u(v[0]), rs, voff, soff, AUX);
    } else if constexpr (C == 2) {
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{f2u(v[0]), f2u(v[1])}, rs, voff, soff, AUX);
    } else {
#pragma unroll
        for (int c = 0; c < C; c += 4) bst128<AUX>(rs, voff + 4 * c, soff, u32x4{f2u(v[c]), f2u(v[c + 1]), f2u(v[c + 2]), f2u(v[c + 3])});
    }
}

template <int C, int AUX = 0>
__device__ __forceinline__ void bst(rsrc_t rs, uint32_t voff, uint32_t soff, const uint8_t *v) {
    if constexpr (C == 1) {
        __builtin_amdgcn_raw_buffer_store_b8(v[0], rs, voff, soff, AUX);
    } else if constexpr (C == 2) {
        __builtin_amdgcn_raw_buffer_store_b16((uint16_t)(v[0] | (v[1] << 8)), rs, voff, soff, AUX);
    } else {
#pragma unroll
        for (int c = 0; c < C; c += 4) {
            const uint32_t w = (uint32_t)v[c] | ((uint32_t)v[c + 1] << 8) | ((uint32_t)v[c + 2] << 16) |
                               ((uint32_t)v[c + 3] << 24);
            __builtin_amdgcn_raw_buffer_store_b32(w, rs, voff + c, soff, AUX);
        }
    }
}

template <int C, int AUX = 0>
__device__ __forceinline__ void bst(rsrc_t rs, uint32_t voff, uint32_t soff, const cf *v) {
    if constexpr (C == 1) {
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{f2u(v[0][0]), f2u(v[0][1])}, rs, voff, soff, AUX);
    } else {
#pragma unroll
        for (int c = 0; c < C; c += 2) bst128<AUX>(rs, voff + 8 * c, soff, u32x4{f2u(v[c][0]), f2u(v[c][1]), f2u(v[c + 1][0]), f2u(v[c + 1][1])});
    }
}

// Raw input words of pass 0: C samples per row.
template <int IN, int C>
struct RawRow;
template <>
struct RawRow<IN_U8, 1> { uint16_t w; };
template <>
struct RawRow<IN_U8, 2> { uint32_t w; };
template <>
struct RawRow<IN_U8, 4> { uint2 w; };
template <>
struct RawRow<IN_U8, 8> { uint4 w; };
template <int C>
struct RawRow<IN_F32, C> { cf w[C]; };

__device__ __forceinline__ float s8f(uint32_t w, int byte) {
    return (float)(int8_t)(uint8_t)(w >> (8 * byte));
}

__device__ __forceinline__ float u8f(uint32_t w, int byte) {
    return (float)(uint8_t)(w >> (8 * byte));
}

// Convert one raw row to C complex samples in integer units (u - 128); the
// 1/256 scale is applied later.  (-1)^n is applied here as a negation of odd
// columns (row strides are even), which the compiler folds into the neg
// modifiers of the first butterflies.
// SIGNED = false leaves the (-1)^n out (the frequency-shifted path folds it into its phasors, the windowed path into its
// weights).  OFFSET = true: the bytes as offset-binary values 0 .. 255 (xormask ^ 0x80808080 turns either byte convention
// into them; v_cvt_f32_ubyteN where the centred form takes the sign-extending conversion).
template <int IN, int C, bool SIGNED = true, bool OFFSET = false>
__device__ __forceinline__ void convert_row(const RawRow<IN, C> &raw, uint32_t xormask, int j0, cf *dst) {
    [[maybe_unused]] auto b2f = [](uint32_t w, int byte) { return OFFSET ? u8f(w, byte) : s8f(w, byte); };
    if constexpr (IN == IN_F32) {
#pragma unroll
        for (int c = 0; c < C; ++c) dst[c] = (SIGNED && ((j0 + c) & 1)) ? -raw.w[c] : raw.w[c];
    } else if constexpr (C == 1) {
        const uint32_t w = (uint32_t)raw.w ^ xormask;
        const cf z = cf{b2f(w, 0), b2f(w, 1)};
        dst[0] = (SIGNED && (j0 & 1)) ? -z : z;
    } else {
        uint32_t words[C / 2];
        if constexpr (C == 2) {
            words[0] = raw.w;
        } else if constexpr (C == 4) {
            words[0] = raw.w.x;
            words[1] = raw.w.y;
        } else {
            words[0] = raw.w.x;
            words[1] = raw.w.y;
            words[2] = raw.w.z;
            words[3] = raw.w.w;
        }
#pragma unroll
        for (int q = 0; q < C / 2; ++q) {
            const uint32_t w = words[q] ^ xormask;
            dst[2 * q] = cf{b2f(w, 0), b2f(w, 1)};         // even column: +
            const cf odd = cf{b2f(w, 2), b2f(w, 3)};
            dst[2 * q + 1] = SIGNED ? -odd : odd;          // odd column: -
        }
    }
}

// ---------------------------------------------------------------------------
// the kernel body
// ---------------------------------------------------------------------------
// MODE_T >= 0 fixes the epilogue at compile time (the hot MAG path); MODE_T = -1
// dispatches on args.mode at run time (uniform branch).
// e^{2 pi i turns}: the argument is reduced to [0, 1) in double; the f64 form is used once per
// lane (prologue), the f32 form once per frame.
__device__ __forceinline__ cf turn_phasor_f64(double turns) {
    turns -= floor(turns);
    double s, c;
    sincospi(2.0 * turns, &s, &c);
    return cf{(float)c, (float)s};
}
__device__ __forceinline__ cf turn_phasor_f32(double turns) {
    turns -= floor(turns);
    float s, c;
    sincospif(2.0f * (float)turns, &s, &c);
    return cf{c, s};
}

// ROT: u8 input is multiplied by a running phasor before the transform (nrf_freq_shifter fused
// into the load, src/nrf.c:843-866): x[n] = (-1)^n (u8/256) e^{2 pi i (phase0 + m delta)}, with
// the shifter's + 0.5 (1+i) restored like the offset-binary DC, analytically in bin N/2.
// RUNS: the 50 %-overlap form (hop == N/2, BASELINE.json's STFT configuration): a workgroup takes RUNS of consecutive
// frames, and the second half of every frame's bytes -- pass-0 rows R0/2 .. R0-1 of each lane -- is kept in registers as
// rows 0 .. R0/2-1 of the next frame, so that every sample is loaded once (FftArgs::run_len frames per run; static units).
// WIN: the taper window fused into pass 0 (fsea_plan_set_window; north_star's "fused unpack+window prologue"): the
// samples x[n] = (u8 - 128) * ((-1)^n w[n]) are never formed -- the first butterfly level of pass 0 takes the weights
// along (dft_regs_win: half a packed op per sample more than without a window), the weights two to a register pair as
// they were loaded.  1 = the lane's P weights are fetched again for every frame (one run of P
// floats per lane from a table that lives in L2; issued in front of the previous frame's row stores, so that they have
// arrived when the frame's bytes are converted and hold registers only from there to there), 2 = they stay in registers
// for the workgroup's lifetime.  The offset-binary DC term is put back behind the last pass from FftArgs::win_dc.
template <class Cfg, int IN, int MODE_T = -1, bool ROT = false, bool RUNS = false, int WIN = 0>
struct FftKernel {
    static_assert(!ROT || IN == IN_U8, "the fused frequency shift is a u8-input path");
    static_assert(WIN == 0 || (IN == IN_U8 && !ROT && Cfg::TWR && (Cfg::OPT & (64 | 1048576 | 8388608)) == 0 && Cfg::P >= 8),
                  "the windowed kernels: u8 input, V1 schedule, register-resident (prescaled) last-pass twiddles");
    static_assert(!RUNS || (IN == IN_U8 && !ROT && Cfg::FPW == 1 && (Cfg::R(0) % 2) == 0), "half-overlap runs: u8 input, one frame per workgroup");
    static constexpr int N = Cfg::N, T = Cfg::T, P = Cfg::P, NP = Cfg::NP, FPW = Cfg::FPW;
    static constexpr int LAST = NP - 1;
    static constexpr int R0 = Cfg::R(0), C0 = Cfg::C(0);
    static constexpr int RL = Cfg::R(LAST), CL = Cfg::C(LAST), NsL = Cfg::Ns(LAST);
    static constexpr bool ONE_WAVE = (Cfg::WG <= 64) || (T <= 64 && (64 % T) == 0);
    // u8 input is transformed in integer units; the 1/256 of x = u8/256.0 is folded
    // into the register-resident last-pass twiddles (and one multiply of the
    // untwiddled row), or applied by the epilogue when those are not in registers.
    static constexpr float SC = (IN == IN_U8) ? (1.0f / 256.0f) : 1.0f;
    static constexpr bool PRESCALED = Cfg::TWR && (IN == IN_U8);
    using Raw = RawRow<IN, C0>;
    // The barrier that separates a pass's LDS reads from the next writes into the same buffer
    // may sit right before those writes (after the butterflies) instead of right after the
    // reads: the reads' latency overlaps the butterflies and the waves' skew is absorbed by them.
    static constexpr bool LAZY_SYNC = (Cfg::OPT & 1) != 0;
    static constexpr bool BATCH_READS = (Cfg::OPT & 2) != 0;
    static constexpr bool TW_HOIST = (Cfg::OPT & 4) != 0;
    static constexpr bool TW_FUSE = (Cfg::OPT & 8) != 0 && (Cfg::ABL & 4) == 0;
    // OPT 4096 / 8192 / 16384: cache policy of the f32 row stores (measurement variants): nt (streaming),
    // sc1 (write through the XCD's L2 and drop the line), sc0 sc1
    static constexpr int ST_AUX = ((Cfg::OPT & 4096) ? 2 : 0) | ((Cfg::OPT & 8192) ? 16 : 0) | ((Cfg::OPT & 16384) ? 1 : 0);
    // One store instruction writes, per frame, the T lanes' CL adjacent elements: T*CL*size bytes in a row.  A
    // streaming (nt) store of less than a 128-byte line reaches HBM as a partial line -- measured write traffic
    // (WRITE_SIZE) of the u8 pixel rows 1.2x (1024 points, 32-byte pieces) to 2.4x (128 points, 16-byte pieces) the
    // row bytes, f32 rows in 64-byte pieces 1.04x -- while the default policy lets L2 assemble the line from the
    // neighbouring rows' pieces first: 128-point DB10 pixels 0.205 -> 0.117 ms per 256 MiB of samples.  So only
    // pieces of a whole line or more are stored nt (profiles/r02_store_policy_by_piece_size.txt).
    // The same goes for a lane whose CL adjacent elements need more than one 16-byte store (four complex bins = 32 bytes:
    // the 1024- and 2048-point layouts in COMPLEX_F32 mode): each of the two instructions then writes every other 16 bytes of
    // the row, half a line at a time -- as nt stores 2.3x slower than the default policy, which lets L2 put the halves
    // together (0.58 -> 0.25 ms per 2^27 samples at 1024 points; found when 1024 moved to four bins per lane in round 3, and
    // 2048 had had it since round 1).
    template <int ELEM_BYTES>
    static constexpr int st_aux() { return (T * CL * ELEM_BYTES >= 128 && CL * ELEM_BYTES <= 16) ? ST_AUX : (ST_AUX & ~2); }
    // OPT 32768: the input loads are streaming (nt) as well
    static constexpr int LD_AUX = (Cfg::OPT & 32768) ? 2 : 0;
    static constexpr bool DEFER = (Cfg::OPT & 128) != 0 && NP == 3;
    static constexpr bool TK_LATE = (Cfg::OPT & 512) != 0 && NP >= 3 && !ONE_WAVE && (Cfg::ABL & 2) == 0;
    static constexpr bool MI = (Cfg::OPT & 256) == 0;  // OPT 256: the +-i butterflies as packed FMAs by (+-1, -+1) (round-1 form)
    // OPT 8388608: the last butterfly level in power form (dft_regs_tw_pw) in the kernels whose rows are powers only:
    // the compile-time MAG / DB10 / DB5 kernels of the V1 schedule with register-resident, fused last-pass twiddles
    static constexpr bool PW = (Cfg::OPT & 8388608) != 0 && (Cfg::OPT & (64 | 1048576)) == 0 && Cfg::TWR && (Cfg::OPT & 8) != 0 &&
                               Cfg::ABL == 0 && RL >= 4 &&
                               (MODE_T == MODE_MAG || MODE_T == MODE_DB10_U8 || MODE_T == MODE_DB5_U8_DCFIX);
    static constexpr bool PX_PACK = (Cfg::OPT & 2097152) != 0;   // pixel epilogue: v_trunc + v_cvt_pk_u8_f32
    static constexpr bool PX_BIAS = (Cfg::OPT & 4194304) != 0;   // ... without the v_trunc (biased round-to-nearest)
    static constexpr bool LANE_ROT = (Cfg::OPT & 16) != 0;       // middle passes
    static constexpr bool LANE_ROT_LAST = (Cfg::OPT & 32) != 0;  // the last pass as well
    // Which frame-lane a physical lane works as in pass I >= 1.  Any bijection is valid: passes meet
    // only through LDS, at logical addresses.  Two layouts need one (scripts/lds_conflicts.py):
    // * 16 bytes per lane (C = 2, P = 32): the pad shifts each 16-lane block by one 16-byte slot and
    //   the ds_read_b128 lane groups mix lanes of two blocks -> rotate block b by b;
    // * 8 bytes per lane with P = 16 (the 4096-point passes): the pad shifts each block by two
    //   8-byte units, so the two blocks of a 32-lane ds_read_b64 group overlap by two units ->
    //   pair block b with block b + T/32 instead of b + 1.
    template <int I>
    static __device__ __forceinline__ int pass_lane(int t) {
        constexpr bool ON = (I == LAST) ? LANE_ROT_LAST : LANE_ROT;
        if constexpr (ON && Cfg::C(I) == 2 && P == 32 && (T % 16) == 0) {
            const int blk = t >> 4;
            return (blk << 4) | ((t - blk) & 15);
        } else if constexpr (ON && Cfg::C(I) == 1 && P == 16 && T == 256) {
            const int blk = t >> 4;
            const int logical = (blk >> 1) | ((blk & 1) << 3);
            return (logical << 4) | (t & 15);
        } else {
            return t;
        }
    }

    static __device__ __forceinline__ void after_reads() {
        if constexpr (BATCH_READS) __builtin_amdgcn_sched_barrier(0);
    }

    // Frames of one workgroup exchange through LDS.  When a frame lives inside
    // a single wavefront no s_barrier is needed: LDS operations of one wave
    // execute in order.
    static __device__ __forceinline__ void frame_sync() {
        if constexpr (Cfg::ABL & 2) {
            return;
        } else if constexpr (ONE_WAVE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
    }

    // the compiler would otherwise hoist the s_barrier above the butterflies (VALU work is not
    // ordered against it), which is the eager placement again
    static __device__ __forceinline__ void lazy_sync() {
        if constexpr (!ONE_WAVE) __builtin_amdgcn_sched_barrier(0);
        frame_sync();
    }

    static constexpr uint32_t IN_BPS = (IN == IN_U8) ? 2 : 8;  // input bytes per complex sample

    // voff: this lane's byte offset inside the unit's window (slot * hop + C0 t samples)
    static __device__ __forceinline__ void load_raw(rsrc_t rs, uint32_t voff, Raw *raw) { load_raw_rows<0, R0>(rs, voff, raw); }

    template <int RLO, int RHI>
    static __device__ __forceinline__ void load_raw_rows(rsrc_t rs, uint32_t voff, Raw *raw) {
        constexpr int STRIDE = N / R0;
        // ABL 512 (measurement only, wrong samples per lane): the frame's bytes fetched 16 per lane, 1 KiB
        // runs per wave instruction -- what a pass-0 layout with 8 adjacent samples per lane would issue
        if constexpr ((Cfg::ABL & 512) != 0 && IN == IN_U8 && C0 == 2 && R0 % 4 == 0 && FPW == 1 && RLO == 0 && RHI == R0) {
#pragma unroll
            for (int r = 0; r < R0 / 4; ++r) {
                const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff * 4u, (uint32_t)(r * 16 * T), LD_AUX);
#pragma unroll
                for (int i = 0; i < 4; ++i) raw[4 * r + i].w = q[i];
            }
            return;
        }
#pragma unroll
        for (int r = RLO; r < RHI; ++r) {
            const uint32_t soff = (uint32_t)(r * STRIDE) * IN_BPS;
            if constexpr (IN == IN_U8) {
                if constexpr (C0 == 1) raw[r].w = __builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, LD_AUX);
                if constexpr (C0 == 2) raw[r].w = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, LD_AUX);
                if constexpr (C0 == 4) {
                    const auto q = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, LD_AUX);
                    raw[r].w.x = q[0];
                    raw[r].w.y = q[1];
                }
                if constexpr (C0 == 8) {
                    const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, LD_AUX);
                    raw[r].w.x = q[0];
                    raw[r].w.y = q[1];
                    raw[r].w.z = q[2];
                    raw[r].w.w = q[3];
                }
            } else if constexpr (C0 == 1) {
                const auto q = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, LD_AUX);
                raw[r].w[0] = cf{u2f(q[0]), u2f(q[1])};
            } else {
#pragma unroll
                for (int c = 0; c < C0; c += 2) {
                    const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 8 * c, soff, LD_AUX);
                    raw[r].w[c] = cf{u2f(q[0]), u2f(q[1])};
                    raw[r].w[c + 1] = cf{u2f(q[2]), u2f(q[3])};
                }
            }
        }
    }

    // twiddle multiply for pass I (I >= 1): v[r*C + c] *= W^{r k}, k = (C t + c) % Ns
    template <int I>
    static __device__ __forceinline__ void apply_twiddles(cf *v, const cf *tw, int t) {
        constexpr int R = Cfg::R(I), C = Cfg::C(I), Ns = Cfg::Ns(I);
        const int k0 = (C * t) % Ns;
#pragma unroll
        for (int r = 1; r < R; ++r) {
            cf w[C];
            if constexpr (Ns % C == 0) {
                ld_c<C>(tw + (r - 1) * Ns + k0, w);
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) w[c] = tw[(r - 1) * Ns + (C * t + c) % Ns];
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if constexpr (Cfg::ABL & 4) v[r * C + c] += w[c];
                else v[r * C + c] = pk_cmul(v[r * C + c], w[c]);
            }
        }
    }

    // LDS addressing.  pad(idx) = idx + 2*(idx / P) is not affine in idx, but every
    // access pattern here is "thread base + r * constant" once the division is taken
    // on the thread part only; written out so that the r-dependent part becomes the
    // immediate offset of the ds_* instruction.
    template <int I>
    static __device__ __forceinline__ void lds_write(cf *lds, const cf *v, int t) {
        constexpr int R = Cfg::R(I), C = Cfg::C(I), Ns = Cfg::Ns(I);
        if constexpr (Cfg::ABL & 2) {
            return;
        } else if constexpr (Ns == 1 && (R % 2) == 0) {
            // column c owns R contiguous outputs at (C t + c) R: the thread's P
            // outputs are the run [P t, P t + P), i.e. pad adds exactly 2 t.
            cf *base = lds + (unsigned)((P + 2) * t);
#pragma unroll
            for (int c = 0; c < C; ++c) {
#pragma unroll
                for (int r = 0; r < R; r += 2) {
                    const cf pr[2] = {v[r * C + c], v[(r + 1) * C + c]};
                    st_c<2>(base + (c * R + r), pr);
                }
            }
        } else if constexpr (Ns % C == 0) {
            const unsigned j = (unsigned)(C * t);
            const unsigned j0 = (j / Ns) * (Ns * R) + (j % Ns);
            if constexpr (Ns % P == 0) {
                cf *base = lds + Cfg::pad(j0);
#pragma unroll
                for (int r = 0; r < R; ++r) st_c<C>(base + r * (Ns + 2 * (Ns / P)), v + r * C);
            } else {
                // Ns < P: rows r = q r' + b share the base of their residue b
                constexpr int Q = P / Ns;
                static_assert(P % Ns == 0 && R % Q == 0, "pad addressing needs Ns | P and (P/Ns) | R");
#pragma unroll
                for (int b = 0; b < Q; ++b) {
                    cf *base = lds + Cfg::pad(j0 + b * Ns);
#pragma unroll
                    for (int rq = 0; rq < R / Q; ++rq) st_c<C>(base + rq * (P + 2), v + (rq * Q + b) * C);
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int j = C * t + c;
                const int j0 = (j / Ns) * (Ns * R) + (j % Ns);
#pragma unroll
                for (int r = 0; r < R; ++r) lds[Cfg::pad(j0 + r * Ns)] = v[r * C + c];
            }
        }
    }

    template <int I>
    static __device__ __forceinline__ void lds_read(const cf *lds, cf *v, int t) {
        constexpr int R = Cfg::R(I), C = Cfg::C(I);
        constexpr int STRIDE = N / R;
        if constexpr (Cfg::ABL & 2) {
            return;
        } else if constexpr (STRIDE % P == 0) {
            const cf *base = lds + Cfg::pad(C * t);
#pragma unroll
            for (int r = 0; r < R; ++r) ld_c<C>(base + r * (STRIDE + 2 * (STRIDE / P)), v + r * C);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) ld_c<C>(lds + Cfg::pad(C * t + r * STRIDE), v + r * C);
        }
    }

    // middle pass I (1 <= I < LAST): read, twiddle, DFT, write back
    // issued(): called once, right behind the LDS reads of pass 1 (OPT 512: the ticket and the next
    // unit's loads are handled there, so that the reads are in flight before the ticket is waited for)
    struct NoHook {
        __device__ __forceinline__ void operator()() const {}
    };
    template <int I, class Hook = NoHook>
    static __device__ __forceinline__ void middle_pass(cf *lds, const cf *lds_all, cf *v, const FftArgs &a, int t0,
                                                       const cf *tw_res = nullptr, Hook &&issued = Hook()) {
        if constexpr (I < LAST) {
            constexpr int R = Cfg::R(I), C = Cfg::C(I);
            const int t = pass_lane<I>(t0);
            const cf *tw = Cfg::TWL ? (lds_all + Cfg::lds_tw_off(I)) : a.tw[I];
            if constexpr (DEFER) {
                lds_read<I>(lds, v, t);
                if constexpr (I == 1) issued();
                after_reads();
                if constexpr (!LAZY_SYNC) frame_sync();
#pragma unroll
                for (int c = 0; c < C; ++c) dft_regs_def<R, C, 1>(v + c, tw_res + c * (R / 2));
            } else if constexpr (TW_HOIST && (Cfg::Ns(I) % C == 0)) {
                constexpr int Ns = Cfg::Ns(I);
                cf w[(R - 1) * C];
                const int k0 = (C * t) % Ns;
#pragma unroll
                for (int r = 1; r < R; ++r) ld_c<C>(tw + (r - 1) * Ns + k0, w + (r - 1) * C);
                lds_read<I>(lds, v, t);
                if constexpr (I == 1) issued();
                after_reads();
                if constexpr (!LAZY_SYNC) frame_sync();
                if constexpr (TW_FUSE) {
#pragma unroll
                    for (int c = 0; c < C; ++c) dft_regs_tw<R, C, C, MI>(v + c, w + c, 1.0f);
                } else {
#pragma unroll
                    for (int r = 1; r < R; ++r) {
#pragma unroll
                        for (int c = 0; c < C; ++c) v[r * C + c] = pk_cmul(v[r * C + c], w[(r - 1) * C + c]);
                    }
                }
            } else {
                lds_read<I>(lds, v, t);
                if constexpr (I == 1) issued();
                after_reads();
                if constexpr (!LAZY_SYNC) frame_sync();  // everyone has read before anyone overwrites
                apply_twiddles<I>(v, tw, t);
            }
            if constexpr (!DEFER && !(TW_FUSE && TW_HOIST && (Cfg::Ns(I) % C == 0))) {
#pragma unroll
                for (int c = 0; c < C; ++c) dft_regs<R, C, (Cfg::ABL & 4) != 0, MI>(v + c);
            }
            if constexpr (LAZY_SYNC) lazy_sync();  // same barrier, after this wave's butterflies
            lds_write<I>(lds, v, t);
            frame_sync();
            middle_pass<I + 1>(lds, lds_all, v, a, t0, tw_res);  // further middle passes: no hook
        }
    }

    static __device__ __forceinline__ uint32_t elem_bytes(int mode) {
        return (mode == MODE_DB10_U8 || mode == MODE_DB5_U8_DCFIX) ? 1u : (mode == MODE_COMPLEX ? 8u : 4u);
    }

    // fused epilogue for the row held as v[r*CL + c] = bin CL t + c + r NsL.
    // out: window of this unit's rows; lane_elem = slot * N + CL * t (first bin of this lane)
    // |v[r CL + c]|^2, or -- PW -- the power the last level left in pp[(r mod RL/2) CL + c] (fsea::dft_power_level)
    static __device__ __forceinline__ float bin_power(const cf *v, [[maybe_unused]] const cf *pp, int r, int c) {
        if constexpr (PW) {
            return pp[(r % (RL / 2)) * CL + c][r / (RL / 2)];
        } else {
            const cf z = v[r * CL + c];
            return __builtin_fmaf(z[0], z[0], z[1] * z[1]);
        }
    }
    // the restored DC term of bin N/2 for the lane and mode that keep it (see epilogue), else 0: what PW kernels hand to
    // the power form of the last level, which has no complex bin to add it to afterwards
    static __device__ __forceinline__ float dc_restore(int mode, int t) {
        const bool patched = (mode == MODE_MAG) || (mode == MODE_DB5_U8_DCFIX);
        return (IN == IN_U8 && !patched && t == 0) ? (PRESCALED ? 0.5f : 128.0f) * (float)N : 0.0f;
    }
    static constexpr bool PW_DC = PW && IN == IN_U8 && MODE_T == MODE_DB10_U8;

    static __device__ __forceinline__ void epilogue(int mode, rsrc_t out, uint32_t lane_elem, cf *v, int t,
                                                    [[maybe_unused]] const cf *pp = nullptr) {
        constexpr float SE = PRESCALED ? 1.0f : SC;  // scale still to apply to re / im
        constexpr float SE2 = SE * SE;
        const bool patched = (mode == MODE_MAG) || (mode == MODE_DB5_U8_DCFIX);
        // Offset-binary input carries a DC term 0.5 per component, which the
        // (-1)^n centring moves to bin N/2 exactly: 0.5 N (1 + i).  The kernel
        // transforms (u - 128) instead and restores that bin analytically in
        // the modes that keep it.
        if (!PW && WIN == 0 && IN == IN_U8 && !patched && t == 0) {
            const float dc = (PRESCALED ? 0.5f : 128.0f) * (float)N;
            v[(RL / 2) * CL] += cf{dc, dc};
        }
        if (mode == MODE_COMPLEX) {
            const uint32_t voff = lane_elem * 8u;
#pragma unroll
            for (int r = 0; r < RL; ++r) {
                cf z[CL];
#pragma unroll
                for (int c = 0; c < CL; ++c) z[c] = v[r * CL + c] * cf{SE, SE};
                bst<CL, st_aux<8>()>(out, voff, (uint32_t)(r * NsL) * 8u, z);
            }
        } else if (mode == MODE_DB10_U8 || mode == MODE_DB5_U8_DCFIX) {
            // 10*log10(p + 1e-20) * s = (10 s log10(2)) * log2(p + 1e-20)
            const float kdb = (mode == MODE_DB10_U8 ? 100.0f : 50.0f) * 0.30102999566398120f;
            const uint32_t voff = lane_elem;
            [[maybe_unused]] uint8_t wide_px[4];
#pragma unroll
            for (int r = 0; r < RL; ++r) {
                const uint32_t soff = (uint32_t)(r * NsL);
                uint8_t px[CL];
                [[maybe_unused]] uint32_t pxw[(CL + 3) / 4] = {};
#pragma unroll
                for (int c = 0; c < CL; ++c) {
                    float p = bin_power(v, pp, r, c);
                    if constexpr (!PRESCALED && IN == IN_U8) p *= SE2;
                    // The reference's "+ 1e-20" only keeps log10 finite: in f32 it changes p by less than
                    // half an ulp whenever the pixel is not clamped to 0 anyway (p > 1e-13), and for
                    // smaller p -- down to log2(0) = -inf, which the conversion saturates -- the pixel
                    // is 0 either way.  Left out: same pixels, one VALU op less per bin.
                    if constexpr (PX_PACK) {
                        // OPT 2097152: truncation (v_trunc_f32), then v_cvt_pk_u8_f32, which saturates to [0, 255] and
                        // drops the byte into place: 2 ops where the cast + clamp + shift/or packing took 2 + ~0.75.
                        // OPT 4194304 (with it): no v_trunc; the conversion rounds to nearest even, so d is lowered by
                        // 0.5 - 2^-25 in the FMA that forms it: floor(d) except for d within ~8e-6 above an odd integer
                        // (one pixel in ~2.5e5 one grey level low; the f32 logarithm itself moves more than that).
                        float d;
                        if constexpr (PX_BIAS) d = __builtin_fmaf(kdb, __builtin_amdgcn_logf(p), -0.49999997f);
                        else d = trunc_f32(kdb * __builtin_amdgcn_logf(p));
                        if constexpr (Cfg::ABL & 128) d = kdb * p;
                        pxw[c / 4] = cvt_pk_u8(d, (uint32_t)(c & 3), pxw[c / 4]);
                    } else {
                        float d = kdb * __builtin_amdgcn_logf(p);
                        if constexpr (Cfg::ABL & 128) d = kdb * p;  // ABL 128 (measurement only): no logarithm
                        int q = (int)d;  // truncation toward zero, as the C cast in the reference
                        q = q < 0 ? 0 : (q > 255 ? 255 : q);
                        px[c] = (uint8_t)q;
                    }
                }
                if constexpr (PX_PACK) {
#pragma unroll
                    for (int c = 0; c < CL; ++c) px[c] = (uint8_t)(pxw[c / 4] >> (8 * (c & 3)));  // (only the patched lanes' byte stores read these)
                }
                if constexpr (Cfg::ABL & 1) {
                    if (px[0] == 255 && px[CL - 1] == 254 && v[0][0] == -1.0f) bst<CL>(out, voff, soff, px);  // (practically) never
                } else if constexpr ((Cfg::ABL & 256) != 0 && CL <= 2 && (RL * CL) % 4 == 0) {
                    // ABL 256 (measurement only, pixels land in the wrong places): four pixels per dword store
                    constexpr int PER = 4 / CL;  // rows per dword
#pragma unroll
                    for (int c = 0; c < CL; ++c) wide_px[(r % PER) * CL + c] = px[c];
                    if (r % PER == PER - 1)
                        bst<4, ST_AUX>(out, lane_elem + (uint32_t)((4 - CL) * t), (uint32_t)((r / PER) * 4 * T), wide_px);
                } else if (patched && r == RL / 2 && t == 0) {
#pragma unroll
                    for (int c = 1; c < CL; ++c) bst<1>(out, voff + c, soff, px + c);
                } else if constexpr (PX_PACK) {
                    if constexpr (CL == 1) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)pxw[0], out, voff, soff, st_aux<1>());
                    else if constexpr (CL == 2) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)pxw[0], out, voff, soff, st_aux<1>());
                    else {
#pragma unroll
                        for (int c = 0; c < CL; c += 4) __builtin_amdgcn_raw_buffer_store_b32(pxw[c / 4], out, voff + c, soff, st_aux<1>());
                    }
                } else {
                    bst<CL, st_aux<1>()>(out, voff, soff, px);
                }
                if (patched && r == RL / 2 - 1 && t == T - 1) bst<1>(out, voff + CL, soff, px + (CL - 1));
            }
        } else if (mode == MODE_DB_F32) {
            // two loops, one transcendental each: written as `mode == MODE_DB_F32 ? log : sqrt` per element the compiler
            // turns the uniform test into v_log_f32 + v_sqrt_f32 + v_cndmask for every bin
            f32_rows<true>(patched, out, lane_elem, v, t, pp);
        } else {
            f32_rows<false>(patched, out, lane_elem, v, t, pp);
        }
    }

    template <bool LOG>
    static __device__ __forceinline__ void f32_rows(bool patched, rsrc_t out, uint32_t lane_elem, cf *v, int t,
                                                    [[maybe_unused]] const cf *pp) {
        constexpr float SE = PRESCALED ? 1.0f : SC;
        constexpr float SE2 = SE * SE;
        const uint32_t voff = lane_elem * 4u;
        [[maybe_unused]] float wide[4];
#pragma unroll
        for (int r = 0; r < RL; ++r) {
            const uint32_t soff = (uint32_t)(r * NsL) * 4u;
            float m[CL];
#pragma unroll
            for (int c = 0; c < CL; ++c) {
                float p = bin_power(v, pp, r, c);
                if constexpr (!PRESCALED && IN == IN_U8) p *= SE2;
                if constexpr (LOG) {
                    m[c] = (10.0f * 0.30102999566398120f) * __builtin_amdgcn_logf(p + 1.0e-20f);
                } else if constexpr (Cfg::ABL & 128) {  // ABL 128 (measurement only): no magnitude arithmetic
                    m[c] = v[r * CL + c][0];
                } else {
                    m[c] = __builtin_amdgcn_sqrtf(p);
                }
            }
            if constexpr (Cfg::ABL & 1) {
                if (m[0] == -1.0f) bst<CL>(out, voff, soff, m);  // never true: sqrt >= 0
            } else if constexpr ((Cfg::ABL & 256) != 0 && CL == 1 && RL % 4 == 0) {
                // ABL 256 (measurement only, bins land in the wrong places): the row stored 16 bytes per
                // lane, 1 KiB runs per wave instruction -- what a last pass with 4 adjacent bins per lane
                // would issue.  The values wait in wide[] until four are there.
                wide[r & 3] = m[0];
                if ((r & 3) == 3) bst<4, ST_AUX>(out, (lane_elem + 3u * (uint32_t)t) * 4u, (uint32_t)((r >> 2) * 4 * T) * 4u, wide);
            } else if (patched && r == RL / 2 && t == 0) {
#pragma unroll
                for (int c = 1; c < CL; ++c) bst<1>(out, voff + 4 * c, soff, m + c);
                // three adjacent dword stores (CL = 4) are merged into ONE buffer_store_dwordx3 by the compiler -- the same
                // hazard as the 16-byte store (tests/test_shipped_artifacts.py found it in two tuning variants): guarded too
                if constexpr (CL >= 4) store_data_guard();
            } else {
                bst<CL, st_aux<4>()>(out, voff, soff, m);
            }
            if (patched && r == RL / 2 - 1 && t == T - 1) bst<1>(out, voff + 4 * CL, soff, m + (CL - 1));
        }
    }

    // ---- dynamic frame distribution ------------------------------------------------
    // Workgroups of one launch do not run at the same speed (measured: the slowest takes
    // 1.5x the fastest, scripts/wg_trace.py), so units (FPW frames) are handed out by
    // atomic ticket counters instead of a static split.  One counter saturates near 90
    // tickets/us on this chip, so there are POOLS of them: the units are cut into POOLS
    // contiguous ranges, a workgroup draws from its home pool (blockIdx % POOLS, which is
    // also its XCD) and steals from the other pools when that one runs dry.  The first
    // unit of a workgroup is static (no atomic on the start-up path) and tickets are
    // requested two iterations ahead, so the atomic's latency is only ever waited for
    // when stealing at the very end of a launch.  Counter words sit in separate 128-byte
    // lines: ctr[32 q] = tickets drawn from pool q, ctr[32 POOLS] = finished workgroups;
    // the last workgroup to finish zeroes them for the next launch on this slot.
    // Sizes whose frames live inside one wavefront have no barrier to publish a ticket with and
    // thousands of independent waves to average over: they keep a static interleave
    // (unit = blockIdx + k * gridDim); the ticket scheme is for the multi-wave sizes.
    static constexpr bool DYNAMIC = !ONE_WAVE && (Cfg::OPT & 2048) == 0;  // OPT 2048 (tuning): static interleave at every size
    static constexpr unsigned POOLS = 8;
    static constexpr unsigned NO_UNIT = 0xffffffffu;

    struct Pools {
        unsigned n_units, grid;
        __device__ __forceinline__ unsigned start(unsigned q) const { return (unsigned)((size_t)n_units * q / POOLS); }
        __device__ __forceinline__ unsigned homed(unsigned q) const { return (grid + POOLS - 1 - q) / POOLS; }
        // unit for ticket t of pool q, or NO_UNIT when the pool is exhausted
        __device__ __forceinline__ unsigned unit(unsigned q, unsigned t) const {
            const unsigned long long idx = (unsigned long long)start(q) + homed(q) + t;
            return idx < start(q + 1) ? (unsigned)idx : NO_UNIT;
        }
    };

    static constexpr bool V2 = (Cfg::OPT & 64) != 0;
    // what a launch does with FftArgs::ctr (host side: which launches need a ticket-counter slot, fsea_api.hip)
    static constexpr int counters_used() {
        if ((Cfg::OPT & 1048576) != 0) return 0;                     // W64: one wave per frame, static interleave
        if (V2 || (Cfg::OPT & 262144) != 0) return 2;
        return DYNAMIC ? 1 : 0;
    }

    static constexpr bool W64 = (Cfg::OPT & 1048576) != 0;

    // ---- taper window (WIN kernels) ----
    static constexpr int WPAIRS = WIN ? P / 2 : 1;
    // LDS of a kernel in complex units: frames, table block, ticket words, and -- windowed kernels -- the DC term's
    // spectrum for the 2 NsL bins around N/2 (FftArgs::win_dc)
    static constexpr int DC_OFF = (Cfg::LDS_ALLOC + 1) & ~1;  // 16-byte aligned
    static constexpr int LDS_CF = WIN ? DC_OFF + 2 * NsL : Cfg::LDS_ALLOC;
    static constexpr int DC_REGS = (2 * NsL + Cfg::WG - 1) / Cfg::WG;
    static_assert(WIN == 0 || (Cfg::TWL || Cfg::TWR), "the DC table rides on the table block's barrier");
    // the lane's P weights, (-1)^n w[n] in the order of its pass-0 registers (FftArgs::win): P/4 16-byte loads, the T
    // lanes' pieces of one load adjacent in memory
    static __device__ __forceinline__ void load_window(rsrc_t rs, int t, cf *wv) {
        const uint32_t voff = (uint32_t)t * 16u;
#pragma unroll
        for (int i = 0; i < P / 4; ++i) {
            const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (uint32_t)(16 * T * i), 0);
            wv[2 * i] = cf{u2f(q[0]), u2f(q[1])};
            wv[2 * i + 1] = cf{u2f(q[2]), u2f(q[3])};
        }
    }
    // pass 0's conversion for the windowed kernels: the byte values as they are (centred, or offset-binary), no (-1)^n --
    // weight and sign are applied by the first butterfly level (pass0_windowed)
    template <bool OFFSET>
    static __device__ __forceinline__ void convert_windowed(const Raw *raw, uint32_t xormask, int t, cf *v) {
#pragma unroll
        for (int r = 0; r < R0; ++r) convert_row<IN, C0, false, OFFSET>(raw[r], OFFSET ? xormask ^ 0x80808080u : xormask, C0 * t, v + r * C0);
    }
    template <int C = 0>
    static __device__ __forceinline__ void pass0_windowed(cf *v, const cf *wv) {
        if constexpr (C < C0) {
            dft_regs_win<R0, C0, C, MI>(v + C, wv);
            pass0_windowed<C + 1>(v, wv);
        }
    }

    static __device__ __forceinline__ void run(const FftArgs &a, cf *lds_all) {
        if constexpr (W64) run_w64(a, lds_all);
        else if constexpr (V2) run_v2(a, lds_all);
        else run_v1(a, lds_all);
    }

    // -----------------------------------------------------------------------------------------
    // W64 schedule (OPT 1048576): N = 64 x 64, one wavefront per frame, 64 points per lane, ONE exchange
    // through LDS and no s_barrier at all.
    //
    // Sample n = 64 n1 + n2, bin k = k1 + 64 k2.  Pass 0: lane n2 transforms x[64 n1 + n2] over n1 (64 points, constant
    // twiddles) -> y[k1]; exchange: element (k1, n2) goes from lane n2 to lane k1; pass 1: lane k1 transforms over n2 with
    // the twiddles W_N^{n2 k1} deferred into the butterflies (dft_regs_def; 32 register pairs per lane, resident) -> k2.
    //   * (-1)^n = (-1)^{n2} is a shift of the spectrum by N/2 bins = of k2 by 32: nothing is negated, row r of the
    //     last pass is bin k1 + 64 (r ^ 32) -- register renaming.
    //   * Loads.  A lane holding one sample per row would load 2 bytes at a time.  Instead lane L loads the dword
    //     sigma(L) + 64 j (j < 32): 256 contiguous bytes per wave instruction, two adjacent samples (n2 = 2d, 2d + 1) of
    //     row n1 = 2j (+1 in lanes 32-63: sigma(L + 32) = sigma(L) + 32).  Lanes L and L + 32 therefore hold the same
    //     two n2 with complementary n1, and ONE v_permlane32_swap_b32 per pair of rows hands each the other's half:
    //     A = perm(keep / send), B = perm(send / keep), swap(A, B) -> A = rows (2j, 2j'), B = rows (2j + 1, 2j' + 1)
    //     of the lane's own n2, the same registers in every lane.  Three VALU ops per four samples.
    //   * Which n2 a lane ends up with is free (passes meet in LDS at logical addresses); sigma and the kept half are
    //     chosen so that the 16 lanes of every ds_write_b64 lane group hold n2 that differ mod 16 (conflict-free):
    //     n2(L) = 2 pi(L & 31) + ((L >> 5) ^ (L & 1)), pi(t) = (t & 16) + ((t & 15) >> 1) + 8 (t & 1).
    //   * LDS: element (k1, n2) at k1 * 66 + n2 (complex units; 16 bytes of pad per row): 64 ds_write_b64 whose lanes
    //     cover one 512-byte row each, 32 ds_read_b128 of the lane's own row (row pitch 33 x 16 bytes: conflict-free).
    //   * Pixel rows (u8 modes): a lane's 64 pixels are bins k1 + 64 r, one byte each.  Four rows are packed into a dword
    //     by v_cvt_pk_u8_f32 (conversion and packing in one op), transposed 4 x 4 inside each quad of lanes (two
    //     v_mov_b32_dpp quad_perm + two v_perm_b32) and stored as dwords: lane 4m + i writes bins 4m .. 4m+3 of row
    //     4q + i, the wave 256 contiguous bytes per instruction, 16 stores per frame.
    // -----------------------------------------------------------------------------------------
    static __device__ __forceinline__ void run_w64(const FftArgs &a, cf *lds) {
        static_assert(!W64 || (N == 4096 && T == 64 && FPW == 1 && NP == 2 && R0 == 64 && RL == 64), "W64 is the 64 x 64 layout");
        static_assert(!W64 || (IN == IN_U8 && !ROT), "W64 serves the u8 kernels");
        static_assert(!W64 || (!Cfg::TWL && !Cfg::TWR), "W64 keeps its (deferred) twiddles in registers: no table block in LDS");
        constexpr int ROW = 66;  // LDS row pitch in complex units
        const int L = threadIdx.x;
        const unsigned b = blockIdx.x;
        const size_t n_units = a.n_frames;
        const int mode = (MODE_T >= 0) ? MODE_T : a.mode;
        const uint32_t xormask = (MODE_T >= 0) ? 0u : a.xormask;
        const uint32_t esz = elem_bytes(mode);
        const size_t total_in = (size_t)IN_BPS * ((a.n_frames - 1) * a.hop + (size_t)N);
        const bool tiled = a.tile_rows != 0;
        const size_t total_out = (size_t)esz * (tiled ? a.out_span : a.n_frames * (size_t)N);
        auto row_elem = [&](size_t f) -> size_t {
            if (!tiled) return f * (size_t)N;
            const uint32_t k = (uint32_t)f / a.tile_rows, y = (uint32_t)f - k * a.tile_rows;
            return (size_t)y * a.pitch_row + (size_t)k * a.pitch_tile;
        };
        // pass-0 identity of this lane
        const int t5 = L & 31, odd = L & 1, hi = L >> 5;
        const int pi = (t5 & 16) + ((t5 & 15) >> 1) + 8 * odd;
        const int n2 = 2 * pi + (hi ^ odd);
        const uint32_t in_voff = 4u * (uint32_t)(pi + 32 * hi);
        const uint32_t sel_a = odd ? 0x07060302u : 0x05040100u;  // the half (sample) that ends up in A: c = L & 1
        const uint32_t sel_b = odd ? 0x05040100u : 0x07060302u;
        // this lane's deferred twiddles of the last pass: row k1 = L of the [64][32] table (build_deferred_table)
        cf twd[RL / 2];
        ld_c<RL / 2>(a.tw_def + (size_t)L * (RL / 2), twd);

        auto load_frame = [&](size_t u, uint32_t *raw) {
            const rsrc_t rs = buffer_window(a.in, (size_t)IN_BPS * u * a.hop, u < n_units ? total_in : 0);
#pragma unroll
            for (int j = 0; j < 32; ++j) raw[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, in_voff, (uint32_t)(256 * j), LD_AUX);
        };
        size_t u = b;
        uint32_t raw[32];
        load_frame(u, raw);
        // the grid size lives in a VGPR: with the 64-point DFT's constants the SGPR file is full, and a gridDim.x re-read
        // from the dispatch packet (s_load_dword) inside the loop is waited for with lgkmcnt(0) -- together with every
        // LDS write in flight
        unsigned grid_v = gridDim.x;
#if defined(__AMDGCN__)
        asm volatile("" : "+v"(grid_v));
#endif
        cf *const wr = lds + n2;               // + ROW k1
        const cf *const rd = lds + ROW * L;    // + n2

        while (u < n_units) {
            cf v[64];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                uint32_t wa = byte_perm(raw[j + 16], raw[j], sel_a);
                uint32_t wb = byte_perm(raw[j + 16], raw[j], sel_b);
                lane_swap32(wa, wb);
                wa ^= xormask;
                wb ^= xormask;
                v[2 * j] = cf{s8f(wa, 0), s8f(wa, 1)};
                v[2 * j + 32] = cf{s8f(wa, 2), s8f(wa, 3)};
                v[2 * j + 1] = cf{s8f(wb, 0), s8f(wb, 1)};
                v[2 * j + 33] = cf{s8f(wb, 2), s8f(wb, 3)};
            }
            dft_regs<64, 1, (Cfg::ABL & 4) != 0, MI>(v);
            frame_sync();  // the previous frame's reads precede these writes
            if constexpr ((Cfg::ABL & 2) == 0) {
#pragma unroll
                for (int k1 = 0; k1 < 64; ++k1) wr[ROW * k1] = v[k1];
            }
            frame_sync();
            const size_t un = u + __builtin_amdgcn_readfirstlane(grid_v);
            if constexpr ((Cfg::ABL & 64) == 0) load_frame(un, raw);
            if constexpr ((Cfg::ABL & 2) == 0) {
                // the first butterflies of pass 1 pair elements j and j + 32: fetched in that order
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    ld_c<2>(rd + 2 * q, v + 2 * q);
                    ld_c<2>(rd + 2 * q + 32, v + 2 * q + 32);
                }
            }
            after_reads();
            if constexpr ((Cfg::ABL & 4) == 0) dft_regs_def<64, 1, 1>(v, twd);
            cf w[64];  // centred order: row r = bin L + 64 r
#pragma unroll
            for (int r = 0; r < 64; ++r) w[r] = v[r ^ 32];
            const rsrc_t out = buffer_window(a.out, (size_t)esz * row_elem(u), total_out);
            if (mode == MODE_DB10_U8 || mode == MODE_DB5_U8_DCFIX) pixels_w64(mode, out, w, L);
            else epilogue(mode, out, (uint32_t)L, w, L);
            u = un;
        }
    }

    static __device__ __forceinline__ void pixels_w64(int mode, rsrc_t out, cf *w, int L) {
        const bool patched = (mode == MODE_DB5_U8_DCFIX);
        if (!patched && L == 0) w[32] += cf{128.0f * (float)N, 128.0f * (float)N};  // DC of the offset-binary samples (see epilogue)
        const float kdb = (mode == MODE_DB10_U8 ? 100.0f : 50.0f) * 0.30102999566398120f;
        const float koff = -16.0f * kdb;  // log2 of the 1/256^2 the integer-unit power still carries
        const uint32_t sel1 = (L & 1) ? 0x03070105u : 0x06020400u;
        const uint32_t sel2 = (L & 2) ? 0x03020706u : 0x05040100u;
        const uint32_t voff = (uint32_t)((L & 3) * 64 + (L & ~3));
        uint32_t left = 0;  // DB5: the pixel of bin N/2 - 1 (row 31, lane 63), copied over bin N/2 (row 32, lane 0)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            uint32_t px = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const cf z = w[4 * q + i];
                const float p = __builtin_fmaf(z[0], z[0], z[1] * z[1]);
                float d = __builtin_fmaf(kdb, __builtin_amdgcn_logf(p), PX_BIAS ? koff - 0.49999997f : koff);
                if constexpr (Cfg::ABL & 128) d = kdb * p;
                px = cvt_pk_u8(PX_BIAS ? d : trunc_f32(d), (uint32_t)i, px);  // truncation toward zero, then saturation: the reference's cast + clamp
            }
            const uint32_t u1 = byte_perm(quad_xor1(px), px, sel1);
            uint32_t rows = byte_perm(quad_xor2(u1), u1, sel2);  // lane 4m + i: bins 4m .. 4m + 3 of row 4q + i
            if (q == 7) left = read_lane(rows, 63) >> 24;
            if (q == 8 && patched && L == 0) rows = (rows & 0xffffff00u) | left;
            if constexpr (Cfg::ABL & 1) {
                if (rows == 0x01020304u && w[0][0] == -1.0f) __builtin_amdgcn_raw_buffer_store_b32(rows, out, voff, (uint32_t)(256 * q), ST_AUX);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(rows, out, voff, (uint32_t)(256 * q), ST_AUX);
            }
        }
    }

    // -----------------------------------------------------------------------------------------
    // V2 schedule for frames spread over several wavefronts (three passes RA x RB x RC).
    //
    // Sample index n = a N/RA + b RC + c, bin k = ka + RA kb + RA RB kc:
    //   pass 0 sums over a (-> ka), twiddle W_{RA RB}^{b ka}, pass 1 over b (-> kb), twiddle
    //   W_N^{c (ka + RA kb)}, pass 2 over c (-> kc).
    // Only the bits a lane holds in registers are forced (a, b, c in turn); which of the other
    // bits are lane bits and which are wave bits is free.  V1 takes them in Stockham order, which
    // makes both exchanges cross-wave: two s_barriers each.  Here the wave bits of passes 0 and 1
    // are the top bits of c, so the first exchange (ka <-> b) stays inside a wavefront -- LDS
    // operations of one wave execute in order, no barrier -- and only the second one crosses
    // waves.  Every element of that second exchange is read by exactly one wave, so the buffer
    // is a partition S_w by reading wave; once wave w has read its S_w nobody else touches it
    // until the next cross-wave write, and w runs its own first exchange of the next frame in
    // it.  Per frame: [pass 0] A-write A-read [pass 1] BARRIER B-write BARRIER B-read [pass 2]:
    // two barriers instead of four, one rendezvous per frame.
    //   * Pass-0 loads: a wave reads 16-byte pieces at 64-byte stride (its c bits), the four
    //     waves of the workgroup cover the lines between them (L1 hits); stores keep 256-byte runs.
    //   * Middle-pass twiddles W_{RA RB}^{b ka} depend on the lane (ka) only: deferred into the
    //     butterflies (dft_regs_def) they are RB/2 register pairs per lane, resident for the
    //     workgroup's lifetime; no twiddle is read from LDS per frame.
    // LDS slots are 16 bytes (two complex, the c0 pair):
    //   A (inside S_w): slot = 65 ka + 16 g + b      writer lane (b, g), reader lane (ka, g)
    //   B:              slot = 17 m + j              m = ka + RA kb, j = c / 2; S_w = rows 64 w ..
    // both conflict-free for ds_write_b128 (8 consecutive lanes -> 8 consecutive slots mod 8) and
    // ds_read_b128 (a 16-lane group -> 16 distinct slots mod 16; 65 and 17 are odd).
    // -----------------------------------------------------------------------------------------
    static __device__ __forceinline__ void wave_order() {
        // LDS operations of one wavefront execute in program order; this only stops the compiler
        // from moving them across (and keeps the CPU emulation's lanes together)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    static __device__ __forceinline__ void run_v2(const FftArgs &a, cf *lds_all) {
        constexpr int RA = Cfg::R(0), RB = Cfg::R(1), RC = Cfg::R(2);
        static_assert(NP == 3 && FPW == 1 && (T % 64) == 0 && !ONE_WAVE, "V2 is for multi-wave frames in three passes");
        static_assert(RA == 16 && RB == 16 && RC == 32 && P == 32, "V2 layout constants are written for 16 x 16 x 32");
        static_assert(Cfg::TWR, "V2 keeps the last pass's twiddles in registers");
        constexpr int G = 64 / RB;            // lane groups per wave (the c bits below the wave bits, above c0)
        constexpr int ROW_B = RC / 2 + 1;     // 16-byte slots per B row (m), odd
        constexpr int SW = 64 * ROW_B;        // slots of one wave's partition S_w
        constexpr int ROW_A = 4 * RB + 1;     // slots between consecutive ka in the A layout, odd
        static_assert((RA - 1) * ROW_A + 16 * (G - 1) + RB <= SW, "the A layout must fit the wave's partition");
        static_assert(2 * SW * (T / 64) <= Cfg::LDS_FRAME, "LDS frame too small for the B layout");

        const int tid = threadIdx.x;
        const int w = tid >> 6, l = tid & 63;
        unsigned *tk = reinterpret_cast<unsigned *>(lds_all + Cfg::LDS_TOTAL);
        cf *lds = lds_all;

        const unsigned bidx = blockIdx.x;
        const bool issuer = (tid == 0);
        const size_t n_units = a.n_frames;
        Pools pools;
        pools.n_units = (unsigned)n_units;
        pools.grid = gridDim.x;
        unsigned cur = bidx % POOLS;

        if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0) {
            const unsigned hw_id = read_hw_id(), xcc_id = read_xcc_id();
            a.trace[32 * bidx + 0] = wall_clock64();
            a.trace[32 * bidx + 2] = __builtin_readcyclecounter();
            a.trace[32 * bidx + 4] = hw_id;
            a.trace[32 * bidx + 5] = xcc_id;
        }

        const int mode = (MODE_T >= 0) ? MODE_T : a.mode;
        const uint32_t xormask = (MODE_T >= 0) ? 0u : a.xormask;
        const uint32_t esz = elem_bytes(mode);
        const size_t total_in = (size_t)IN_BPS * ((a.n_frames - 1) * a.hop + (size_t)N);
        const bool tiled = a.tile_rows != 0;
        const size_t total_out = (size_t)esz * (tiled ? a.out_span : a.n_frames * (size_t)N);
        auto row_elem = [&](size_t f) -> size_t {  // element offset of frame f's row (FPW == 1 here)
            if (!tiled) return f * (size_t)N;
            const uint32_t k = (uint32_t)f / a.tile_rows, y = (uint32_t)f - k * a.tile_rows;
            return (size_t)y * a.pitch_row + (size_t)k * a.pitch_tile;
        };
        // pass 0: lane (b, g) of wave w holds samples n = a N/RA + b RC + (w G + g) C0 + c0
        const int b0 = l % RB, g = l / RB;
        const int n_base = b0 * RC + (w * G + g) * C0;
        // ABL 16 (measurement only, wrong results): the V1 load mapping, 256-byte runs per wave
        const uint32_t in_voff = (uint32_t)IN_BPS * (uint32_t)((Cfg::ABL & 16) ? C0 * tid : n_base);
        // ABL 8 (measurement only): static unit interleave, next unit's bytes requested right after the
        // conversion of this one's
        constexpr bool STATIC = (Cfg::ABL & 8) != 0;
        // pass 1: lane (ka, g); pass 2: thread m = ka + RA kb = tid
        const int ka = l % RA;
        const int m = tid;
        const uint32_t out_elem = (uint32_t)m;

        size_t u = (size_t)pools.start(cur) + bidx / POOLS;
        if (u >= pools.start(cur + 1)) u = n_units;
        if constexpr (STATIC) u = bidx;
        unsigned tick_next = 0;

        constexpr int TAB_COPY = Cfg::TAB_SMALL;
        constexpr int TAB_REGS = (TAB_COPY + Cfg::WG - 1) / Cfg::WG;
        cf tabv[TAB_REGS];
#pragma unroll
        for (int i = 0; i < TAB_REGS; ++i) {
            const int e = tid + i * Cfg::WG;
            tabv[i] = a.tw_small[e < TAB_COPY ? e : TAB_COPY - 1];
        }
        // this lane's deferred middle-pass twiddles: RB/2 pairs, resident
        cf tw1[RB / 2];
        ld_c<RB / 2>(a.tw_def + ka * (RB / 2), tw1);
        Raw raw[R0];
        load_raw(buffer_window(a.in, (size_t)IN_BPS * u * a.hop, u < n_units ? total_in : 0), in_voff, raw);
        if (!STATIC && issuer) tick_next = atomicAdd(a.ctr + 32 * cur, 1u);

#pragma unroll
        for (int i = 0; i < TAB_REGS; ++i) {
            const int e = tid + i * Cfg::WG;
            lds_all[Cfg::LDS_FRAME + (e < TAB_COPY ? e : TAB_COPY - 1)] = tabv[i];
        }
        __syncthreads();
        cf twl[(RL - 1) * CL];
        {
            const cf *hi = lds_all + Cfg::LDS_HI, *lo = lds_all + Cfg::LDS_LO;
#pragma unroll
            for (int r = 1; r < RL; ++r) {
                const unsigned e = (unsigned)r * (unsigned)m;
                cf tw = pk_cmul(hi[e >> 6], lo[e & 63u]);
                if constexpr (PRESCALED) tw = tw * cf{SC, SC};
                twl[r - 1] = tw;
            }
        }

        cf ebase[ROT ? C0 : 1];
        if constexpr (ROT) {
#pragma unroll
            for (int c = 0; c < C0; ++c) {
                const unsigned n0 = (unsigned)(n_base + c);
                const cf e = turn_phasor_f64(a.rot_delta * (double)n0);
                ebase[c] = (n0 & 1u) ? -e : e;
            }
        }

        unsigned par = 0;
        if (!STATIC && u >= n_units) {
            if (issuer) {
                unsigned nu = pools.unit(cur, tick_next);
                for (unsigned k = 1; nu == NO_UNIT && k < POOLS; ++k) {
                    const unsigned q = (cur + k) % POOLS;
                    nu = pools.unit(q, atomicAdd(a.ctr + 32 * q, 1u));
                    if (nu != NO_UNIT) cur = q;
                }
                tk[0] = nu;
                tick_next = atomicAdd(a.ctr + 32 * cur, 1u);
            }
            __syncthreads();
            const unsigned nu = __builtin_amdgcn_readfirstlane(tk[0]);
            __syncthreads();
            u = (nu == NO_UNIT) ? n_units : (size_t)nu;
            load_raw(buffer_window(a.in, (size_t)IN_BPS * u * a.hop, u < n_units ? total_in : 0), in_voff, raw);
        }

        // LDS addresses (complex units; a slot is two complex)
        cf *const sw = lds + 2 * SW * w;                                   // this wave's partition
        cf *const a_wr = sw + 2 * (16 * g + b0);                           // + 2 ROW_A ka
        const cf *const a_rd = sw + 2 * (ROW_A * ka + 16 * g);             // + 2 b
        cf *const b_wr = lds + 2 * (ROW_B * ka + (w * G + g));             // + 2 ROW_B RA kb
        const cf *const b_rd = lds + 2 * ROW_B * m;                        // + 2 j

        unsigned iter = 0;
        if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0) a.trace[32 * bidx + 6] = wall_clock64();
        while (u < n_units) {
            if (!STATIC && issuer) {
                unsigned nu = pools.unit(cur, tick_next);
                for (unsigned k = 1; nu == NO_UNIT && k < POOLS; ++k) {
                    const unsigned q = (cur + k) % POOLS;
                    nu = pools.unit(q, atomicAdd(a.ctr + 32 * q, 1u));
                    if (nu != NO_UNIT) cur = q;
                }
                tk[par] = nu;
                tick_next = atomicAdd(a.ctr + 32 * cur, 1u);
            }

            cf v[P];
            if constexpr (ROT) {
                const size_t first = u * a.hop;
                const cf ef = turn_phasor_f32(a.rot_phase0 + a.rot_delta * (double)first);
                cf fr[C0];
#pragma unroll
                for (int c = 0; c < C0; ++c) fr[c] = pk_cmul(ebase[c], ef);
#pragma unroll
                for (int r = 0; r < R0; ++r) {
                    convert_row<IN, C0, false>(raw[r], a.xormask, n_base, v + r * C0);
                    const cf wr = a.rot_row[r];
#pragma unroll
                    for (int c = 0; c < C0; ++c) {
                        v[r * C0 + c] = pk_cmul(v[r * C0 + c] + cf{128.0f, 128.0f}, pk_cmul_uniform(fr[c], wr));
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R0; ++r) convert_row<IN, C0>(raw[r], xormask, n_base, v + r * C0);
            }
            size_t un = u + gridDim.x;
            if constexpr (STATIC) {
                load_raw(buffer_window(a.in, (size_t)IN_BPS * un * a.hop, un < n_units ? total_in : 0), in_voff, raw);
            }
#pragma unroll
            for (int c = 0; c < C0; ++c) dft_regs<R0, C0>(v + c);

            // exchange A, inside this wave's partition: ka <-> b
            if constexpr ((Cfg::ABL & 32) == 0) {  // ABL 32 (measurement only): no exchange A
                wave_order();  // this wave's reads of the previous frame (B) precede these writes
#pragma unroll
                for (int r = 0; r < RA; ++r) st_c<2>(a_wr + 2 * ROW_A * r, v + 2 * r);
                wave_order();
#pragma unroll
                for (int r = 0; r < RB; ++r) ld_c<2>(a_rd + 2 * r, v + 2 * r);
                after_reads();
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) dft_regs_def<RB, 2, 1>(v + c, tw1);

            __syncthreads();  // every wave has read its partition: the cross-wave writes may land
            if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0 && iter == 0) a.trace[32 * bidx + 7] = wall_clock64();
            // exchange B, across waves: row m = ka + RA kb gets this lane's c pair at j = w G + g
#pragma unroll
            for (int r = 0; r < RB; ++r) st_c<2>(b_wr + 2 * ROW_B * RA * r, v + 2 * r);
            __syncthreads();
            // nothing but the writes sits between the two barriers; the ticket word is read in
            // front of the data (LDS returns in order), and the next unit's bytes are requested
            // while the rows come in
            const unsigned tkv = STATIC ? 0u : tk[par];
#pragma unroll
            for (int j = 0; j < RC / 2; ++j) ld_c<2>(b_rd + 2 * j, v + 2 * j);
            if constexpr (!STATIC) {
                __builtin_amdgcn_sched_barrier(0);  // the reads are issued before the ticket is waited for
                const unsigned nu = __builtin_amdgcn_readfirstlane(tkv);
                par ^= 1u;
                un = (nu == NO_UNIT) ? n_units : (size_t)nu;
                load_raw(buffer_window(a.in, (size_t)IN_BPS * un * a.hop, un < n_units ? total_in : 0), in_voff, raw);
            }
            after_reads();

            if constexpr (TW_FUSE) {
                dft_regs_tw<RL, 1, 1>(v, twl, PRESCALED ? SC : 1.0f);
            } else {
                if constexpr (PRESCALED) v[0] = v[0] * cf{SC, SC};
#pragma unroll
                for (int r = 1; r < RL; ++r) v[r] = pk_cmul(v[r], twl[r - 1]);
                dft_regs<RL, 1>(v);
            }
            epilogue(mode, buffer_window(a.out, (size_t)esz * row_elem(u), total_out), out_elem, v, m);
            u = un;
            if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0 && iter < 24) a.trace[32 * bidx + 8 + iter] = wall_clock64();
            ++iter;
        }

        if (!STATIC && issuer) {
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned finished = atomicAdd(a.ctr + 32 * POOLS, 1u);
            if (finished == gridDim.x - 1) {
#pragma unroll
                for (unsigned q = 0; q <= POOLS; ++q) a.ctr[32 * q] = 0;
            }
        }
        if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0) {
            a.trace[32 * bidx + 1] = wall_clock64();
            a.trace[32 * bidx + 3] = __builtin_readcyclecounter();
        }
    }

    static __device__ __forceinline__ void run_v1(const FftArgs &a, cf *lds_all) {
        // OPT 1024: static priority for the second resident workgroup of every CU (blocks are placed
        // round-robin, so block b and b + grid/2 share a CU): one of the two co-resident waves of a
        // SIMD always wins the VALU, the other fills its gaps
        if constexpr ((Cfg::OPT & 1024) != 0) {
            if (blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_setprio(1);
        }
        const int tid = threadIdx.x;
        const int slot = (FPW == 1) ? 0 : tid / T;  // frame index inside the unit = LDS region
        const int t = (FPW == 1) ? tid : tid % T;
        cf *lds = lds_all + slot * Cfg::LDS_FRAME;
        unsigned *tk = reinterpret_cast<unsigned *>(lds_all + Cfg::LDS_TOTAL);  // 2 words: next unit, ping-pong

        const unsigned b = blockIdx.x;
        const bool issuer = (tid == 0);
        const size_t n_units = RUNS ? (a.n_frames + a.run_len - 1) / a.run_len : (a.n_frames + FPW - 1) / FPW;
        Pools pools;
        pools.n_units = (unsigned)n_units;
        pools.grid = gridDim.x;
        unsigned cur = b % POOLS;  // pool this workgroup is drawing from (issuer lane only)

        if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0) {
            const unsigned hw_id = read_hw_id(), xcc_id = read_xcc_id();
            a.trace[32 * b + 0] = wall_clock64();
            a.trace[32 * b + 2] = __builtin_readcyclecounter();
            a.trace[32 * b + 4] = hw_id;
            a.trace[32 * b + 5] = xcc_id;
        }

        const int mode = (MODE_T >= 0) ? MODE_T : a.mode;
        // the compile-time-mode kernels (MAG, DB5, DB10) are the raw-int8 (HackRF, flip) path: their
        // bytes are the signed samples already, no XOR; offset-binary input goes through the
        // run-time-mode kernel
        const uint32_t xormask = (MODE_T >= 0) ? 0u : a.xormask;
        const uint32_t esz = elem_bytes(mode);
        const size_t total_in = (size_t)IN_BPS * ((a.n_frames - 1) * a.hop + (size_t)N);
        const bool tiled = a.tile_rows != 0;
        const size_t total_out = (size_t)esz * (tiled ? a.out_span : a.n_frames * (size_t)N);
        const uint32_t in_voff = (uint32_t)IN_BPS * ((uint32_t)slot * (uint32_t)a.hop + (uint32_t)(C0 * t));
        const int tl = pass_lane<LAST>(t);  // this lane's place in the last pass
        const uint32_t out_elem = (uint32_t)slot * (tiled ? a.pitch_row : (uint32_t)N) + (uint32_t)(CL * tl);
        // element offset of frame f's row
        auto row_elem = [&](size_t f) -> size_t {
            if (!tiled) return f * (size_t)N;
            const uint32_t k = (uint32_t)f / a.tile_rows, y = (uint32_t)f - k * a.tile_rows;
            return (size_t)y * a.pitch_row + (size_t)k * a.pitch_tile;
        };

        // Prologue: every independent request is issued before the first wait, so that
        // the latencies overlap: the ticket for the second unit, unit 0's bytes (HBM
        // starts streaming at once), the register-resident last-pass twiddles, then the
        // middle-pass tables for LDS.
        size_t u = b;  // static interleave (single-wave frames; short launches of the multi-wave sizes)
        unsigned tick_next = 0;
        // The ticket pools pay for themselves when a workgroup gets many units (they even out the unequal progress of
        // workgroups and XCDs); with a handful each, the plain interleave is faster -- no atomics, no ticket word to wait
        // for, nothing to steal at the end (profiles/r02_static_vs_ticket_distribution.txt).  The host decides per launch.
        const bool dyn = !RUNS && DYNAMIC && a.dynamic_units != 0;
        // half-overlap runs: the frame this workgroup is on, and its place in the run
        [[maybe_unused]] size_t fcur = RUNS ? (size_t)b * a.run_len : 0;
        [[maybe_unused]] unsigned jpos = 0;
        if (dyn) {
            u = (size_t)pools.start(cur) + b / POOLS;         // static first unit
            if (u >= pools.start(cur + 1)) u = n_units;       // more workgroups than units in this pool
        }
        // the table block's loads go first: they mostly hit L2 / the Infinity Cache, and loads
        // return in order, so behind unit 0's bytes (HBM) they would only be usable when those are
        constexpr int TAB_COPY = (Cfg::TWL || Cfg::TWR) ? Cfg::TAB_SMALL : 0;
        constexpr int TAB_REGS = (TAB_COPY + Cfg::WG - 1) / Cfg::WG;
        cf tabv[TAB_REGS > 0 ? TAB_REGS : 1];
#pragma unroll
        for (int i = 0; i < TAB_REGS; ++i) {
            const int e = tid + i * Cfg::WG;
            tabv[i] = a.tw_small[e < TAB_COPY ? e : TAB_COPY - 1];  // clamped, not predicated: no branch
        }
        // deferred middle-pass twiddles (OPT 128): row k = (C1 t + c) % Ns1 of the table, per column
        constexpr int R1 = Cfg::R(1), C1 = Cfg::C(1), Ns1 = Cfg::Ns(1);
        cf tw1[DEFER ? C1 * (R1 / 2) : 1];
        if constexpr (DEFER) {
            const int t1 = pass_lane<1>(t);
#pragma unroll
            for (int c = 0; c < C1; ++c) ld_c<R1 / 2>(a.tw_def + ((C1 * t1 + c) % Ns1) * (R1 / 2), tw1 + c * (R1 / 2));
        }
        Raw raw[R0];
        load_raw(buffer_window(a.in, (size_t)IN_BPS * (RUNS ? fcur : u * FPW) * a.hop, u < n_units ? total_in : 0), in_voff, raw);
        if (dyn) {
            if (issuer) tick_next = atomicAdd(a.ctr + 32 * cur, 1u);  // ticket for the second unit
        }
        // taper window: this lane's weights, and the DC term's spectrum in the band [N/2 - NsL, N/2 + NsL) -- the rows
        // RL/2 - 1 and RL/2 of the last pass -- on its way to LDS (held in registers it costs spills at 16384 points)
        [[maybe_unused]] const rsrc_t win_rs = buffer_window(WIN ? a.win : nullptr, 0, WIN ? (size_t)N * 4u : 0);
        cf wv[WPAIRS];
        cf dcv[WIN ? DC_REGS : 1];
        if constexpr (WIN != 0) {
            load_window(win_rs, t, wv);
#pragma unroll
            for (int i = 0; i < DC_REGS; ++i) {
                const int e = tid + i * Cfg::WG;
                dcv[i] = a.win_dc[e < 2 * NsL ? e : 2 * NsL - 1];  // clamped, as the table block above
            }
        }

        // The small twiddle block (middle-pass tables + HI/LO factors, a few KiB) goes to LDS, and
        // the last pass's register-resident twiddles W^{r k}, k = CL t + c, are built from the
        // two factor tables: W^{m} = HI[m >> 6] * LO[m & 63].  (Loading those (RL-1) CL twiddles
        // per lane from the full table instead cost 32 MB of L2 traffic per launch and a
        // 5.8 us prologue.)  All of it overlaps the latency of unit 0's bytes requested above.
        if constexpr (TAB_COPY > 0) {
#pragma unroll
            for (int i = 0; i < TAB_REGS; ++i) {
                const int e = tid + i * Cfg::WG;
                // lanes past the end rewrite the last entry with its own value (loaded clamped above)
                lds_all[FPW * Cfg::LDS_FRAME + (e < TAB_COPY ? e : TAB_COPY - 1)] = tabv[i];
            }
            if constexpr (WIN != 0) {
#pragma unroll
                for (int i = 0; i < DC_REGS; ++i) {
                    const int e = tid + i * Cfg::WG;
                    lds_all[DC_OFF + (e < 2 * NsL ? e : 2 * NsL - 1)] = dcv[i];
                }
            }
            if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0) a.trace[32 * b + 28] = wall_clock64();  // tables arrived
            __syncthreads();
            if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0) a.trace[32 * b + 29] = wall_clock64();  // tables in LDS
        }
        // The last pass's register twiddles W^{r k}, k = CL tl + c, r < RL.  Each could be one product of two gathered
        // factors, HI[m >> 6] * LO[m & 63], m = r k: 2 (RL-1) CL LDS gathers per lane, which at 8192 points keep the LDS
        // pipe of a CU busy for 1.5 us in front of the first frame (scripts/wg_trace.py: tables in LDS at 1.6 us, twiddles
        // built at 3.1 us).  Instead r = RB a + b: W^{r k} = W^{RB a k} * W^{b k}, so only RL/RB - 1 + RB - 1 values are
        // gathered (10 instead of 31 at RL = 32) and the rest are products of two of those -- the same number of complex
        // multiplies, one rounding more on the composite ones.  OPT 524288 (tuning): the direct form.
        constexpr bool TWL_DIRECT = (Cfg::OPT & 524288) != 0;
        cf twl[Cfg::TWR ? (RL - 1) * CL : 1];
        if constexpr (Cfg::TWR) {
            const cf *hi = lds_all + Cfg::LDS_HI, *lo = lds_all + Cfg::LDS_LO;
            auto root = [&](unsigned m) -> cf { return pk_cmul(hi[m >> 6], lo[m & 63u]); };  // W^m, m < N * RL
            constexpr int RB = TWL_DIRECT ? RL : (RL >= 16 ? 8 : (RL >= 4 ? 2 : RL));  // low digit of r
            constexpr int RA = RL / RB;
#pragma unroll
            for (int c = 0; c < CL; ++c) {
                const unsigned k = (unsigned)(CL * tl + c);
                cf wb[RB], wa[RA];
#pragma unroll
                for (int bb = 1; bb < RB; ++bb) wb[bb] = root((unsigned)bb * k);
#pragma unroll
                for (int aa = 1; aa < RA; ++aa) wa[aa] = root((unsigned)(RB * aa) * k);
#pragma unroll
                for (int r = 1; r < RL; ++r) {
                    const int aa = r / RB, bb = r % RB;
                    cf w = aa == 0 ? wb[bb] : (bb == 0 ? wa[aa] : pk_cmul(wa[aa], wb[bb]));
                    if constexpr (PRESCALED) w = w * cf{SC, SC};
                    twl[(r - 1) * CL + c] = w;
                }
            }
        }

        // Fused frequency shift: this lane's samples are n = C0 t + c + r N/R0, so their phasors
        // factor as (frame phase) x ebase[c] x rot_row[r]; ebase also carries (-1)^n (N/R0 is even).
        cf ebase[ROT ? C0 : 1];
        if constexpr (ROT) {
#pragma unroll
            for (int c = 0; c < C0; ++c) {
                const unsigned n0 = (unsigned)(C0 * t + c);
                const cf e = turn_phasor_f64(a.rot_delta * (double)n0);
                ebase[c] = (n0 & 1u) ? -e : e;
            }
        }

        // A workgroup whose static unit does not exist still has to look for work (another
        // pool may be long): resolve its first ticket synchronously.
        unsigned par = 0;
        if (dyn && u >= n_units) {
            if (issuer) {
                unsigned nu = pools.unit(cur, tick_next);
                for (unsigned k = 1; nu == NO_UNIT && k < POOLS; ++k) {
                    const unsigned q = (cur + k) % POOLS;
                    nu = pools.unit(q, atomicAdd(a.ctr + 32 * q, 1u));
                    if (nu != NO_UNIT) cur = q;
                }
                tk[0] = nu;
                tick_next = atomicAdd(a.ctr + 32 * cur, 1u);
            }
            __syncthreads();
            const unsigned nu = __builtin_amdgcn_readfirstlane(tk[0]);
            __syncthreads();
            u = (nu == NO_UNIT) ? n_units : (size_t)nu;
            load_raw(buffer_window(a.in, (size_t)IN_BPS * (u * FPW) * a.hop, u < n_units ? total_in : 0), in_voff, raw);
        }

        unsigned iter = 0;
        if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0) a.trace[32 * b + 6] = wall_clock64();  // prologue done
        while (u < n_units) {
            // Next unit: the ticket requested one iteration ago has long arrived.  If the
            // pool it came from is exhausted, steal from the others (synchronous; this only
            // happens at the end of a launch).  Published to the workgroup by the first
            // barrier of this iteration.
            if (dyn && issuer) {
                unsigned nu = pools.unit(cur, tick_next);
                for (unsigned k = 1; nu == NO_UNIT && k < POOLS; ++k) {
                    const unsigned q = (cur + k) % POOLS;
                    nu = pools.unit(q, atomicAdd(a.ctr + 32 * q, 1u));
                    if (nu != NO_UNIT) cur = q;
                }
                tk[par] = nu;
                tick_next = atomicAdd(a.ctr + 32 * cur, 1u);
            }

            // OPT 262144 (tuning): the two workgroups of a CU (blocks b and b + grid / 2) tell each other how many frames they
            // have done; the one behind raises its issue priority, the one ahead lowers it
            [[maybe_unused]] unsigned partner_progress = 0;
            if constexpr ((Cfg::OPT & 262144) != 0) {
                unsigned *prog = a.ctr + 9 * 32;
                const unsigned half = gridDim.x >> 1;
                const unsigned partner = b < half ? b + half : b - half;
                if (tid == 0) __hip_atomic_store(prog + b, iter + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                partner_progress = __hip_atomic_load(prog + partner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }

            // A lane without a frame (ragged last unit) still runs the barriers; it simply
            // transforms the zeros its loads returned and its stores are dropped.
            cf v[P];
            if constexpr (ROT) {
                // phase of the frame's first sample: reduced in double, evaluated in float (a common
                // factor of the whole frame, so its rounding cannot disturb the spectrum's shape)
                const size_t first = (u * FPW + (size_t)slot) * a.hop;
                const cf ef = turn_phasor_f32(a.rot_phase0 + a.rot_delta * (double)first);
                cf fr[C0];
#pragma unroll
                for (int c = 0; c < C0; ++c) fr[c] = pk_cmul(ebase[c], ef);
#pragma unroll
                for (int r = 0; r < R0; ++r) {
                    convert_row<IN, C0, false>(raw[r], a.xormask, C0 * t, v + r * C0);
                    const cf wr = a.rot_row[r];
#pragma unroll
                    for (int c = 0; c < C0; ++c) {
                        // u8 = (u8 - 128) + 128: the shifter rotates the offset-binary value itself
                        v[r * C0 + c] = pk_cmul(v[r * C0 + c] + cf{128.0f, 128.0f}, pk_cmul_uniform(fr[c], wr));
                    }
                }
            } else if constexpr (WIN != 0) {
                if (a.win_offset != 0) convert_windowed<true>(raw, xormask, t, v);
                else convert_windowed<false>(raw, xormask, t, v);
            } else {
#pragma unroll
                for (int r = 0; r < R0; ++r) convert_row<IN, C0>(raw[r], xormask, C0 * t, v + r * C0);
            }
            if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0 && iter == 0) {
                // the first unit's bytes have arrived (one converted value pinned in front of the stamp)
                asm volatile("" ::"v"(v[0]));
                a.trace[32 * b + 30] = wall_clock64();
            }
            if constexpr (WIN != 0) {
                pass0_windowed(v, wv);
            } else {
#pragma unroll
                for (int c = 0; c < C0; ++c) dft_regs<R0, C0, (Cfg::ABL & 4) != 0, MI>(v + c);
            }
            if constexpr (LAZY_SYNC) lazy_sync();  // the previous frame's last read is complete everywhere
            lds_write<0>(lds, v, t);
            frame_sync();
            if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0 && iter == 0) a.trace[32 * b + 7] = wall_clock64();  // first pass 0 done
            // prefetch: the next unit is known to every lane now; its bytes stay in flight
            // during the rest of the transform
            size_t un = u + gridDim.x;
            [[maybe_unused]] size_t fnext = 0;
            [[maybe_unused]] bool same_run = false;
            if constexpr (RUNS) {
                static_assert(!RUNS || !TK_LATE, "half-overlap runs use the plain prefetch");
                same_run = (jpos + 1 < a.run_len) && (fcur + 1 < a.n_frames);
                un = same_run ? u : u + gridDim.x;
                fnext = same_run ? fcur + 1 : un * (size_t)a.run_len;
            }
            if constexpr (TK_LATE) {
                // the ticket word is read in front of pass 1's data (LDS returns in order) and only
                // waited for once those reads are in flight
                const unsigned tkv = tk[par];
                middle_pass<1>(lds, lds_all, v, a, t, tw1, [&]() {
                    __builtin_amdgcn_sched_barrier(0);
                    if (dyn) {
                        const unsigned nu = __builtin_amdgcn_readfirstlane(tkv);
                        par ^= 1u;
                        un = (nu == NO_UNIT) ? n_units : (size_t)nu;
                    }
                    load_raw(buffer_window(a.in, (size_t)IN_BPS * (un * FPW) * a.hop, un < n_units ? total_in : 0), in_voff,
                             raw);
                });
            } else {
                if constexpr ((Cfg::OPT & 65536) != 0) {
                    if ((iter + (blockIdx.x >= gridDim.x / 2 ? 1u : 0u)) & 1u) __builtin_amdgcn_s_setprio(2);
                    else __builtin_amdgcn_s_setprio(0);
                }
                if constexpr ((Cfg::OPT & 262144) != 0) {
                    const unsigned pp = __builtin_amdgcn_readfirstlane(partner_progress);
                    if (pp > iter + 1u) __builtin_amdgcn_s_setprio(3);
                    else if (pp < iter + 1u) __builtin_amdgcn_s_setprio(0);
                    else __builtin_amdgcn_s_setprio(1);
                }
                if (dyn) {
                    if constexpr (Cfg::ABL & 2) __syncthreads();  // the ablation removed the barrier that publishes tk
                    const unsigned nu = __builtin_amdgcn_readfirstlane(tk[par]);
                    par ^= 1u;
                    un = (nu == NO_UNIT) ? n_units : (size_t)nu;
                    // OPT 65536 (tuning): the two workgroups of a CU take turns at the higher issue priority, frame by frame;
                    // OPT 131072 (tuning): a workgroup that has had fewer units than its pool's average so far raises its
                    // priority, one that is ahead lowers it (the unit number says how many the pool has handed out)
                    if constexpr ((Cfg::OPT & 131072) != 0) {
                        if (nu != NO_UNIT) {
                            const unsigned q = (unsigned)(((size_t)nu * POOLS) / pools.n_units);  // pool the unit came from
                            const unsigned handed = nu - pools.start(q < POOLS ? q : POOLS - 1);
                            const unsigned mine = (iter + 2u) * pools.homed(b % POOLS);         // units this workgroup has had, scaled
                            if (mine < handed) __builtin_amdgcn_s_setprio(3);
                            else __builtin_amdgcn_s_setprio(0);
                        }
                    }
                }
                if constexpr (RUNS) {
                    const rsrc_t rs = buffer_window(a.in, (size_t)IN_BPS * fnext * a.hop, un < n_units ? total_in : 0);
                    if (same_run) {  // the bytes of rows R0/2 .. R0-1 are rows 0 .. R0/2-1 of the next frame: only its second half is fetched
#pragma unroll
                        for (int r = 0; r < R0 / 2; ++r) raw[r] = raw[r + R0 / 2];
                        load_raw_rows<R0 / 2, R0>(rs, in_voff, raw);
                    } else {
                        load_raw(rs, in_voff, raw);
                    }
                } else if constexpr ((Cfg::ABL & 64) == 0) {  // ABL 64 (measurement only): the first unit's bytes are reused
                    load_raw(buffer_window(a.in, (size_t)IN_BPS * (un * FPW) * a.hop, un < n_units ? total_in : 0), in_voff, raw);
                }
                middle_pass<1>(lds, lds_all, v, a, t, tw1);
            }

            // last pass
            lds_read<LAST>(lds, v, tl);
            after_reads();
            if constexpr (!LAZY_SYNC) frame_sync();  // the buffer is free for the next frame's pass 0
            [[maybe_unused]] cf pp[PW ? (RL / 2) * CL : 1];
            if constexpr (PW) {
                const float dc_hi = PW_DC ? dc_restore(mode, tl) : 0.0f;
                dft_regs_tw_pw<RL, CL, CL, MI, PW_DC>(v, twl, PRESCALED ? SC : 1.0f, pp, dc_hi);
#pragma unroll
                for (int c = 1; c < CL; ++c) dft_regs_tw_pw<RL, CL, CL, MI, false>(v + c, twl + c, PRESCALED ? SC : 1.0f, pp + c);
            } else if constexpr (Cfg::TWR && TW_FUSE) {
#pragma unroll
                for (int c = 0; c < CL; ++c) dft_regs_tw<RL, CL, CL, MI>(v + c, twl + c, PRESCALED ? SC : 1.0f);
            } else if constexpr (Cfg::TWR) {
                if constexpr (PRESCALED) {
#pragma unroll
                    for (int c = 0; c < CL; ++c) v[c] = v[c] * cf{SC, SC};  // row 0 has no twiddle
                }
#pragma unroll
                for (int r = 1; r < RL; ++r) {
#pragma unroll
                    for (int c = 0; c < CL; ++c) {
                        if constexpr (Cfg::ABL & 4) v[r * CL + c] += twl[(r - 1) * CL + c];
                        else v[r * CL + c] = pk_cmul(v[r * CL + c], twl[(r - 1) * CL + c]);
                    }
                }
            } else {
                apply_twiddles<LAST>(v, a.tw[LAST], tl);
            }
            if constexpr (!(Cfg::TWR && TW_FUSE)) {
#pragma unroll
                for (int c = 0; c < CL; ++c) dft_regs<RL, CL, (Cfg::ABL & 4) != 0, MI>(v + c);
            }
            if constexpr (WIN != 0) {
                // the offset-binary DC term's spectrum, for this lane's bins of the two rows around N/2
                // (two bins at a time: the 2 CL values in flight at once cost spills at 1024 points, CL = 4)
                constexpr int CB = CL >= 2 ? 2 : 1;
#pragma unroll
                for (int c = 0; c < CL; c += CB) {
                    cf dc_lo[CB], dc_hi[CB];
                    ld_c<CB>(lds_all + DC_OFF + CL * tl + c, dc_lo);
                    ld_c<CB>(lds_all + DC_OFF + NsL + CL * tl + c, dc_hi);
#pragma unroll
                    for (int j = 0; j < CB; ++j) {
                        v[(RL / 2 - 1) * CL + c + j] += dc_lo[j];
                        v[(RL / 2) * CL + c + j] += dc_hi[j];
                    }
                }
                if constexpr (WIN == 1) {
                    // the next frame's weights: requested in front of this frame's row stores (loads return in order, and a
                    // wait for a load issued behind the stores would wait for those as well)
                    load_window(win_rs, t, wv);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            epilogue(mode, buffer_window(a.out, (size_t)esz * row_elem(RUNS ? fcur : u * FPW), total_out), out_elem, v, tl, pp);
            u = un;
            if constexpr (RUNS) {
                fcur = fnext;
                jpos = same_run ? jpos + 1 : 0;
            }
            if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0 && iter < 24) a.trace[32 * b + 8 + iter] = wall_clock64();
            ++iter;
        }

        if constexpr ((Cfg::OPT & 262144) != 0) {
            if (tid == 0) __hip_atomic_store(a.ctr + 9 * 32 + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // clean for the next launch
        }
        // this worker is done: its outstanding ticket request must have landed before it is
        // counted, so that the last worker's reset cannot be overtaken by a late increment
        if (dyn && issuer) {
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned finished = atomicAdd(a.ctr + 32 * POOLS, 1u);
            if (finished == gridDim.x - 1) {
#pragma unroll
                for (unsigned q = 0; q <= POOLS; ++q) a.ctr[32 * q] = 0;
            }
        }
        if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0) {
            a.trace[32 * b + 1] = wall_clock64();
            a.trace[32 * b + 3] = __builtin_readcyclecounter();
        }
    }
};

}  // namespace fsea
