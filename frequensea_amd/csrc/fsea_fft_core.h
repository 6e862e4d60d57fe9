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
#include "fsea_opt.h"     // opt::..., abl::...
#ifdef FSEA_TUNE
#include <fsea_pk_asm_tune.h>  // cross-lane primitives of the tuning library's W64 schedule
#endif

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
template <int L, int K>
__device__ __forceinline__ void bfly_const(cf &a, cf &b) {
    if constexpr (K == 0) {
        const cf t = b;
        b = a - t;
        a = a + t;
    } else if constexpr (4 * K == L) {  // w = -i : w b = (b.y, -b.x): packed adds with a swizzle and a one-lane negation
        const cf t = b;
        b = pk_add_pi(a, t);
        a = pk_add_mi(a, t);
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
template <int R, int HALF, int BASE = 0, int K = 0>
__device__ __forceinline__ void dft_const_level(cf *y) {
    if constexpr (BASE < R) {
        bfly_const<2 * HALF, K>(y[BASE + K], y[BASE + K + HALF]);
        if constexpr (K + 1 < HALF) dft_const_level<R, HALF, BASE, K + 1>(y);
        else dft_const_level<R, HALF, BASE + 2 * HALF, 0>(y);
    }
}
template <int R, int HALF>
__device__ __forceinline__ void dft_const_levels(cf *y) {
    if constexpr (HALF < R) {
        dft_const_level<R, HALF>(y);
        dft_const_levels<R, 2 * HALF>(y);
    }
}

// R-point DFT (forward, -1 exponent) over x[0], x[S], ..., x[(R-1)S];
// natural order in and out.  Bit reversal is register renaming only.
template <int R, int S, bool SKIP = false>
__device__ __forceinline__ void dft_regs(cf *x) {
    if constexpr (SKIP) return;
    static_assert(R >= 2 && R <= 64 && (R & (R - 1)) == 0, "radix must be 2..64");
    constexpr int BITS = ilog2c(R);
    cf y[R];
#pragma unroll
    for (int i = 0; i < R; ++i) y[bitrev_c(i, BITS)] = x[i * S];
    dft_const_levels<R, 1>(y);
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
template <int R, int S, int E0>
__device__ __forceinline__ void dft_regs_win(cf *x, const cf *wv) {
    static_assert(R >= 2 && R <= 64 && (R & (R - 1)) == 0, "radix must be 2..64");
    cf y[R];
    dft_win_level<R, S, E0>(x, wv, y);
    dft_const_levels<R, 2>(y);
#pragma unroll
    for (int i = 0; i < R; ++i) x[i * S] = y[i];
}

// The same DFT with the inter-pass twiddle multiply x[i] *= tw[(i - 1) TS] (i >= 1) fused into
// the first butterfly level, whose own twiddle is 1: with a = x[i] tw_i and b = x[i + R/2] tw_j,
//   plus = a + b = pk_cmul_add(x[i + R/2], tw_j, a)   2 pk_fma
//   minus = 2 a - plus                                  1 pk_fma
// i.e. 5 packed ops per pair instead of 6 (3 instead of 4 for the untwiddled row 0).
// SC0: scale applied to row 0 (the other rows carry it in their twiddles), 1 = none.
template <int R, int S, int TS>
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
    dft_const_levels<R, 2>(y);
#pragma unroll
    for (int i = 0; i < R; ++i) x[i * S] = y[i];
}

// The same twiddled DFT with the inter-pass twiddles deferred into the butterfly levels
// (polynomial form): with x_i scaled by om^i (om = W^k, one value per lane), y_k = P(om W_R^k),
// P(z) = sum x_i z^i = E(z^2) + z O(z^2).  The level that merges sub-transforms of size `half`
// then uses the twiddles om^{R/(2 half)} W_{2 half}^j, j < half, of which the upper half is -i
// times the lower one: R/2 run-time twiddles per lane instead of R-1, every butterfly the
// general 3-op form (plus = a + w b in two packed FMAs, minus = 2a - plus in one) -- the same
// 3 R log2 R / 2 packed ops as dft_regs_tw.
// tw layout: [om^{R/2}] [om^{R/4}] [om^{R/8} W_8^{0,1}] [om^{R/16} W_16^{0..3}] ... (fsea_tables.h).
// DROP (measurement only, wrong results; abl::DROP_16_OPS / DROP_32_OPS): the levels with HALF <= DROP leave out the `minus` FMA.
template <int R, int HALF, int TS, int DROP = 0>
__device__ __forceinline__ void dft_def_level(cf *y, const cf *tw) {
    if constexpr (HALF < R) {
        constexpr int OFF = HALF / 2, DISTINCT = HALF >= 2 ? HALF / 2 : 1;
#pragma unroll
        for (int base = 0; base < R; base += 2 * HALF) {
#pragma unroll
            for (int k = 0; k < DISTINCT; ++k) {
                cf &a = y[base + k], &b = y[base + k + HALF];
                const cf plus = pk_cmul_add(b, tw[(OFF + k) * TS], a);
                if constexpr (HALF <= DROP) b = a;          // (no op: the old a, a value of its own, so that nothing downstream folds)
                else b = cf_fma(a, cf{2.0f, 2.0f}, -plus);
                a = plus;
            }
            if constexpr (HALF >= 2) {
#pragma unroll
                for (int k = 0; k < DISTINCT; ++k) {
                    cf &a = y[base + DISTINCT + k], &b = y[base + DISTINCT + k + HALF];
                    const cf plus = pk_cmul_add_mi(b, tw[(OFF + k) * TS], a);
                    if constexpr (HALF <= DROP) b = a;
                    else b = cf_fma(a, cf{2.0f, 2.0f}, -plus);
                    a = plus;
                }
            }
        }
        dft_def_level<R, 2 * HALF, TS, DROP>(y, tw);
    }
}

template <int R, int S, int TS, int DROP = 0>
__device__ __forceinline__ void dft_regs_def(cf *x, const cf *tw) {
    static_assert(R >= 4 && R <= 64 && (R & (R - 1)) == 0, "radix must be 4..64");
    constexpr int BITS = ilog2c(R);
    cf y[R];
#pragma unroll
    for (int i = 0; i < R; ++i) y[bitrev_c(i, BITS)] = x[i * S];
    dft_def_level<R, 1, TS, DROP>(y, tw);
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
#ifndef FSEA_STATIC_CONTIG
#define FSEA_STATIC_CONTIG 0
#endif
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
    // OPT: schedule options, names and meanings in fsea_opt.h (all give identical results); the product configurations and what
    // each one uses: fsea_configs.h.  ABL: measurement-only ablations of the tuning library (abl::..., results wrong by design).
    static constexpr int OPT = OPT_;
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
    static_assert(WIN == 0 || (Cfg::TWR && (Cfg::OPT & opt::TUNE_ONLY) == 0 && Cfg::P >= 8),
                  "the windowed kernels: V1 schedule, register-resident last-pass twiddles");
    static_assert(WIN == 0 || !RUNS || (IN == IN_U8 && !ROT), "windowed half-overlap runs: the plain u8 kernel only");
    // which windowed kernels put the DC term 0.5 (1 + i) D[k] back from FftArgs::win_dc: the u8 kernels (the offset-binary
    // bytes' own DC) and the frequency-shifted ones (the shifter's + 0.5 (1 + i), src/nrf.c:857-858); f32-complex input has none
    static constexpr bool WIN_DC = WIN != 0 && IN == IN_U8;
    // f32-complex input with a taper: the next frame's rows (64 VGPRs of prefetch at 32 points per lane) are requested behind
    // the row stores instead of behind pass 0 -- with the 32 resident weights on top they would not fit, and this kernel is
    // the NUT_BUFFER_F64 branch of nrf_fft_process (src/nrf.c:607-612): one frame per call, nothing to prefetch
    static constexpr bool LATE_LOAD = WIN != 0 && IN == IN_F32;
    static constexpr bool DC_IN_REGS = WIN_DC && ((Cfg::OPT & opt::WIN_DC_REGS) != 0 ||
                                                  ((Cfg::OPT & opt::WIN_DC_REGS_MAG) != 0 && MODE_T == MODE_MAG && !ROT));  // see DC_REGS below
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
    static constexpr bool BATCH_READS = (Cfg::OPT & opt::BATCH_READS) != 0;
    static constexpr bool TW_HOIST = (Cfg::OPT & opt::TW_HOIST) != 0;
    static constexpr bool TW_FUSE = (Cfg::OPT & opt::TW_FUSE) != 0 && (Cfg::ABL & abl::NO_FLOPS) == 0;
    // opt::ST_NT: the row stores are streaming (nt) -- where one store covers a whole line, st_aux() below
    static constexpr int ST_AUX = (Cfg::OPT & opt::ST_NT) ? 2 : 0;
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
    // opt::LD_NT: the input loads are streaming (nt) as well
    static constexpr int LD_AUX = (Cfg::OPT & opt::LD_NT) ? 2 : 0;
    static constexpr bool DEFER = (Cfg::OPT & opt::DEFER) != 0 && NP == 3;
    static constexpr bool PX_PACK = (Cfg::OPT & opt::PX_PACK) != 0;   // pixel epilogue: v_trunc + v_cvt_pk_u8_f32
    static constexpr bool PX_BIAS = (Cfg::OPT & opt::PX_BIAS) != 0;   // ... without the v_trunc (biased round-to-nearest)
    static constexpr bool LANE_ROT = (Cfg::OPT & opt::LANE_ROT) != 0;       // middle passes
    static constexpr bool LANE_ROT_LAST = (Cfg::OPT & opt::LANE_ROT_LAST) != 0;  // the last pass as well
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
        if constexpr (Cfg::ABL & abl::NO_LDS) {
            return;
        } else if constexpr (ONE_WAVE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
    }

    static constexpr uint32_t IN_BPS = (IN == IN_U8) ? 2 : 8;  // input bytes per complex sample

    // voff: this lane's byte offset inside the unit's window (slot * hop + C0 t samples)
    static __device__ __forceinline__ void load_raw(rsrc_t rs, uint32_t voff, Raw *raw) { load_raw_rows<0, R0>(rs, voff, raw); }

    template <int RLO, int RHI>
    static __device__ __forceinline__ void load_raw_rows(rsrc_t rs, uint32_t voff, Raw *raw) {
        constexpr int STRIDE = N / R0;
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
                if constexpr (Cfg::ABL & abl::NO_FLOPS) v[r * C + c] += w[c];
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
        if constexpr (Cfg::ABL & abl::NO_LDS) {
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
        if constexpr (Cfg::ABL & abl::NO_LDS) {
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
    template <int I>
    static __device__ __forceinline__ void middle_pass(cf *lds, const cf *lds_all, cf *v, const FftArgs &a, int t0,
                                                       const cf *tw_res = nullptr) {
        if constexpr (I < LAST) {
            constexpr int R = Cfg::R(I), C = Cfg::C(I);
            const int t = pass_lane<I>(t0);
            const cf *tw = Cfg::TWL ? (lds_all + Cfg::lds_tw_off(I)) : a.tw[I];
            if constexpr (DEFER) {
                lds_read<I>(lds, v, t);
                after_reads();
                frame_sync();
#pragma unroll
                for (int c = 0; c < C; ++c) dft_regs_def<R, C, 1, (Cfg::ABL & abl::DROP_32_OPS) ? 2 : ((Cfg::ABL & abl::DROP_16_OPS) ? 1 : 0)>(v + c, tw_res + c * (R / 2));
            } else if constexpr (TW_HOIST && (Cfg::Ns(I) % C == 0)) {
                constexpr int Ns = Cfg::Ns(I);
                cf w[(R - 1) * C];
                const int k0 = (C * t) % Ns;
#pragma unroll
                for (int r = 1; r < R; ++r) ld_c<C>(tw + (r - 1) * Ns + k0, w + (r - 1) * C);
                lds_read<I>(lds, v, t);
                after_reads();
                frame_sync();
                if constexpr (TW_FUSE) {
#pragma unroll
                    for (int c = 0; c < C; ++c) dft_regs_tw<R, C, C>(v + c, w + c, 1.0f);
                } else {
#pragma unroll
                    for (int r = 1; r < R; ++r) {
#pragma unroll
                        for (int c = 0; c < C; ++c) v[r * C + c] = pk_cmul(v[r * C + c], w[(r - 1) * C + c]);
                    }
                }
            } else {
                lds_read<I>(lds, v, t);
                after_reads();
                frame_sync();  // everyone has read before anyone overwrites
                apply_twiddles<I>(v, tw, t);
            }
            if constexpr (!DEFER && !(TW_FUSE && TW_HOIST && (Cfg::Ns(I) % C == 0))) {
#pragma unroll
                for (int c = 0; c < C; ++c) dft_regs<R, C, (Cfg::ABL & abl::NO_FLOPS) != 0>(v + c);
            }
            lds_write<I>(lds, v, t);
            frame_sync();
            middle_pass<I + 1>(lds, lds_all, v, a, t0, tw_res);
        }
    }

    static __device__ __forceinline__ uint32_t elem_bytes(int mode) {
        return (mode == MODE_DB10_U8 || mode == MODE_DB5_U8_DCFIX) ? 1u : (mode == MODE_COMPLEX ? 8u : 4u);
    }

    // fused epilogue for the row held as v[r*CL + c] = bin CL t + c + r NsL.
    // out: window of this unit's rows; lane_elem = slot * N + CL * t (first bin of this lane)
    // |v[r CL + c]|^2
    static __device__ __forceinline__ float bin_power(const cf *v, int r, int c) {
        const cf z = v[r * CL + c];
        return __builtin_fmaf(z[0], z[0], z[1] * z[1]);
    }

    static __device__ __forceinline__ void epilogue(int mode, rsrc_t out, uint32_t lane_elem, cf *v, int t) {
        constexpr float SE = PRESCALED ? 1.0f : SC;  // scale still to apply to re / im
        constexpr float SE2 = SE * SE;
        const bool patched = (mode == MODE_MAG) || (mode == MODE_DB5_U8_DCFIX);
        // Offset-binary input carries a DC term 0.5 per component, which the
        // (-1)^n centring moves to bin N/2 exactly: 0.5 N (1 + i).  The kernel
        // transforms (u - 128) instead and restores that bin analytically in
        // the modes that keep it.
        if (WIN == 0 && IN == IN_U8 && !patched && t == 0) {
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
#pragma unroll
            for (int r = 0; r < RL; ++r) {
                const uint32_t soff = (uint32_t)(r * NsL);
                uint8_t px[CL];
                [[maybe_unused]] uint32_t pxw[(CL + 3) / 4] = {};
#pragma unroll
                for (int c = 0; c < CL; ++c) {
                    float p = bin_power(v, r, c);
                    if constexpr (!PRESCALED && IN == IN_U8) p *= SE2;
                    // The reference's "+ 1e-20" only keeps log10 finite: in f32 it changes p by less than
                    // half an ulp whenever the pixel is not clamped to 0 anyway (p > 1e-13), and for
                    // smaller p -- down to log2(0) = -inf, which the conversion saturates -- the pixel
                    // is 0 either way.  Left out: same pixels, one VALU op less per bin.
                    if constexpr (PX_PACK) {
                        // opt::PX_PACK: truncation (v_trunc_f32), then v_cvt_pk_u8_f32, which saturates to [0, 255] and
                        // drops the byte into place: 2 ops where the cast + clamp + shift/or packing took 2 + ~0.75.
                        // opt::PX_BIAS (with it): no v_trunc; the conversion rounds to nearest even, so d is lowered by
                        // 0.5 - 2^-25 in the FMA that forms it: floor(d) except for d within ~8e-6 above an odd integer
                        // (one pixel in ~2.5e5 one grey level low; the f32 logarithm itself moves more than that).
                        float d;
                        if constexpr (PX_BIAS) d = __builtin_fmaf(kdb, __builtin_amdgcn_logf(p), -0.49999997f);
                        else d = trunc_f32(kdb * __builtin_amdgcn_logf(p));
                        if constexpr (Cfg::ABL & abl::NO_EPILOGUE_MATH) d = kdb * p;
                        pxw[c / 4] = cvt_pk_u8(d, (uint32_t)(c & 3), pxw[c / 4]);
                    } else {
                        float d = kdb * __builtin_amdgcn_logf(p);
                        if constexpr (Cfg::ABL & abl::NO_EPILOGUE_MATH) d = kdb * p;  // ABL 128 (measurement only): no logarithm
                        int q = (int)d;  // truncation toward zero, as the C cast in the reference
                        q = q < 0 ? 0 : (q > 255 ? 255 : q);
                        px[c] = (uint8_t)q;
                    }
                }
                if constexpr (PX_PACK) {
#pragma unroll
                    for (int c = 0; c < CL; ++c) px[c] = (uint8_t)(pxw[c / 4] >> (8 * (c & 3)));  // (only the patched lanes' byte stores read these)
                }
                if constexpr (Cfg::ABL & abl::NO_STORES) {
                    if (px[0] == 255 && px[CL - 1] == 254 && v[0][0] == -1.0f) bst<CL>(out, voff, soff, px);  // (practically) never
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
            f32_rows<true>(patched, out, lane_elem, v, t);
        } else {
            f32_rows<false>(patched, out, lane_elem, v, t);
        }
    }

    template <bool LOG>
    static __device__ __forceinline__ void f32_rows(bool patched, rsrc_t out, uint32_t lane_elem, cf *v, int t) {
        constexpr float SE = PRESCALED ? 1.0f : SC;
        constexpr float SE2 = SE * SE;
        const uint32_t voff = lane_elem * 4u;
#pragma unroll
        for (int r = 0; r < RL; ++r) {
            const uint32_t soff = (uint32_t)(r * NsL) * 4u;
            float m[CL];
#pragma unroll
            for (int c = 0; c < CL; ++c) {
                float p = bin_power(v, r, c);
                if constexpr (!PRESCALED && IN == IN_U8) p *= SE2;
                if constexpr (LOG) {
                    m[c] = (10.0f * 0.30102999566398120f) * __builtin_amdgcn_logf(p + 1.0e-20f);
                } else if constexpr (Cfg::ABL & abl::NO_EPILOGUE_MATH) {  // ABL 128 (measurement only): no magnitude arithmetic
                    m[c] = v[r * CL + c][0];
                } else {
                    m[c] = __builtin_amdgcn_sqrtf(p);
                }
            }
            if constexpr (Cfg::ABL & abl::NO_STORES) {
                if (m[0] == -1.0f) bst<CL>(out, voff, soff, m);  // never true: sqrt >= 0
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
    static constexpr bool DYNAMIC = !ONE_WAVE;
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

    // The two schedules of the tuning library (fsea_fft_tune_members.h); a product build has neither.
#ifdef FSEA_TUNE
    static constexpr bool V2 = (Cfg::OPT & opt::V2) != 0, W64 = (Cfg::OPT & opt::W64) != 0;
#else
    static constexpr bool V2 = false, W64 = false;
    static_assert((Cfg::OPT & opt::TUNE_ONLY) == 0 && Cfg::ABL == 0, "tuning-only option or ablation in a product build");
#endif
    // what a launch does with FftArgs::ctr (host side: which launches need a ticket-counter slot, fsea_api.hip):
    // 0 = nothing, 1 = ticket pools when FftArgs::dynamic_units, 2 = ticket pools always (V2)
    static constexpr int counters_used() { return W64 ? 0 : (V2 ? 2 : (DYNAMIC ? 1 : 0)); }

    // ---- taper window (WIN kernels) ----
    static constexpr int WPAIRS = WIN ? P / 2 : 1;
    // LDS of a kernel in complex units: frames, table block, ticket words, and -- windowed kernels -- the DC term's
    // spectrum for the 2 NsL bins around N/2 (FftArgs::win_dc)
    static constexpr int DC_OFF = (Cfg::LDS_ALLOC + 1) & ~1;  // 16-byte aligned
    static constexpr int LDS_CF = (WIN_DC && !DC_IN_REGS) ? DC_OFF + 2 * NsL : Cfg::LDS_ALLOC;
    static constexpr int DC_REGS = (2 * NsL + Cfg::WG - 1) / Cfg::WG;
    // Where a lane's share of the DC table lives.  Its 2 CL values never change from frame to frame; read from LDS per frame
    // (round 4) they cost two ds_read + a wait right in front of the row stores -- 1.0 of the 2.0 us a Hann taper added to
    // the 49 us headline launch (profiles/r05_window_prologue.txt, build 3).  Where the register budget has room for them
    // (opt::WIN_DC_REGS in the size's configuration; opt::WIN_DC_REGS_MAG: its compile-time MAG kernels only, at 1024 in the
    // 8 x 16 x 8 layout, 2048 and 16384, where the other kinds spill with 2 CL more register pairs; 4096 keeps the LDS form)
    // they are loaded once, straight from the table, and stay (DC_IN_REGS).
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
    // The frame loop is software-pipelined: frame k+1's bytes are requested behind frame k's pass 0 and its row stores are issued
    // behind them, so at the top of an iteration the wave has [loads of this frame, stores of the previous one] in flight, in
    // that order, and needs only the loads.  gfx950 counts both in vmcnt (they retire in order), and the compiler merges the
    // loop header's two predecessors by aligning their most recent operations: coming from the prologue there are the loads
    // alone, so it concludes "at most L - 1 operations may be outstanding" -- and in the steady state that wait also covers
    // EVERY ROW STORE OF THE PREVIOUS FRAME (s_waitcnt vmcnt(15) ... vmcnt(0) where vmcnt(47) ... vmcnt(32) would do): a frame
    // starts only after the previous frame's stores have been acknowledged.  With opt::BALANCE_PX / BALANCE_MAG the prologue
    // issues as many stores as an iteration does, through a zero-sized buffer window (the range check drops them: no traffic),
    // so that both predecessors present the same profile and the compiler's own count leaves the stores out of the wait.
    // Worth 2-7 % on the compile-time pixel kernels at every size (byte stores are acknowledged late) and 2-4 % on long
    // launches of the MAG kernels at 1024 ... 8192 points (1 % in the half-overlap form); nothing at 16384 points or on the
    // run-time-mode kernels, and the windowed kernels' longer prologue loses it again (profiles/r05_prologue_stores.txt) --
    // hence per mode and per configuration.
    static constexpr bool BALANCE = WIN == 0 && !LATE_LOAD && !ROT && IN == IN_U8 &&
                                    (((MODE_T == MODE_DB5_U8_DCFIX || MODE_T == MODE_DB10_U8) && (Cfg::OPT & opt::BALANCE_PX) != 0) ||
                                     (MODE_T == MODE_MAG && (Cfg::OPT & opt::BALANCE_MAG) != 0));
    static constexpr int BALANCE_STORES = 32;
    static __device__ __forceinline__ void balance_vmcnt() {
        if constexpr (BALANCE) {
            const rsrc_t nowhere = buffer_window(nullptr, 0, 0);
#pragma unroll
            for (int i = 0; i < BALANCE_STORES; ++i) {
                // distinct, non-adjacent offsets: identical stores would be eliminated, adjacent ones merged into wider ones
                __builtin_amdgcn_raw_buffer_store_b32(0u, nowhere, (uint32_t)(64 * i), 0u, 0);
            }
        }
    }
    template <bool OFFSET>
    static __device__ __forceinline__ void convert_windowed(const Raw *raw, uint32_t xormask, int t, cf *v) {
#pragma unroll
        for (int r = 0; r < R0; ++r) convert_row<IN, C0, false, OFFSET>(raw[r], OFFSET ? xormask ^ 0x80808080u : xormask, C0 * t, v + r * C0);
    }
    template <int C = 0>
    static __device__ __forceinline__ void pass0_windowed(cf *v, const cf *wv) {
        if constexpr (C < C0) {
            dft_regs_win<R0, C0, C>(v + C, wv);
            pass0_windowed<C + 1>(v, wv);
        }
    }

#ifdef FSEA_TUNE
#include "fsea_fft_tune_members.h"
    static __device__ __forceinline__ void run(const FftArgs &a, cf *lds_all) {
        if constexpr (W64) run_w64(a, lds_all);
        else if constexpr (V2) run_v2(a, lds_all);
        else run_v1(a, lds_all);
    }
#else
    static __device__ __forceinline__ void run(const FftArgs &a, cf *lds_all) { run_v1(a, lds_all); }
#endif

    static __device__ __forceinline__ void run_v1(const FftArgs &a, cf *lds_all) {
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
#if FSEA_STATIC_CONTIG  // measurement only: each workgroup takes a contiguous run of units instead (profiles/r05_static_contiguous.txt)
        [[maybe_unused]] const size_t contig_per = (n_units + gridDim.x - 1) / gridDim.x;
        [[maybe_unused]] size_t contig_end = n_units;
        if constexpr (!RUNS) {
            u = (size_t)b * contig_per;
            contig_end = u + contig_per < n_units ? u + contig_per : n_units;
            if (u >= n_units) u = n_units;
        }
#endif
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
        // deferred middle-pass twiddles (opt::DEFER): row k = (C1 t + c) % Ns1 of the table, per column
        constexpr int R1 = Cfg::R(1), C1 = Cfg::C(1), Ns1 = Cfg::Ns(1);
        cf tw1[DEFER ? C1 * (R1 / 2) : 1];
        if constexpr (DEFER) {
            const int t1 = pass_lane<1>(t);
#pragma unroll
            for (int c = 0; c < C1; ++c) ld_c<R1 / 2>(a.tw_def + ((C1 * t1 + c) % Ns1) * (R1 / 2), tw1 + c * (R1 / 2));
        }
        // taper window: this lane's weights, and the DC term's spectrum in the band [N/2 - NsL, N/2 + NsL) -- the rows
        // RL/2 - 1 and RL/2 of the last pass -- on its way to LDS (held in registers it costs spills at 16384 points).
        // Requested in front of unit 0's bytes: they come from L2 and must not queue behind the HBM burst of the launch's start.
        [[maybe_unused]] const rsrc_t win_rs = buffer_window(WIN ? a.win : nullptr, 0, WIN ? (size_t)N * 4u : 0);
        cf wv[WPAIRS];
        cf dcv[(WIN_DC && !DC_IN_REGS) ? DC_REGS : 1];
        cf dcr[DC_IN_REGS ? 2 * CL : 1];  // this lane's bins of the two last-pass rows around N/2: [c] row RL/2 - 1, [CL + c] row RL/2
        if constexpr (DC_IN_REGS) {
#pragma unroll
            for (int c = 0; c < CL; ++c) {
                dcr[c] = a.win_dc[CL * tl + c];
                dcr[CL + c] = a.win_dc[NsL + CL * tl + c];
            }
        }
        if constexpr (WIN != 0) load_window(win_rs, t, wv);
        if constexpr (WIN_DC && !DC_IN_REGS) {
#pragma unroll
            for (int i = 0; i < DC_REGS; ++i) {
                const int e = tid + i * Cfg::WG;
                dcv[i] = a.win_dc[e < 2 * NsL ? e : 2 * NsL - 1];  // clamped, as the table block above
            }
        }
        // the ticket for the second unit goes out IN FRONT of unit 0's bytes: in the loop the ticket request is older than the
        // frame's loads and stores, and the prologue has to present the same order, or the compiler's merged wait for the
        // ticket (issuer wave, top of every iteration of a long launch) is vmcnt(0) -- all row stores of the previous frame
        // acknowledged -- where the ticket alone would do (balance_vmcnt above)
        if (dyn) {
            if (issuer) tick_next = atomicAdd(a.ctr + 32 * cur, 1u);
        }
        Raw raw[R0];
        load_raw(buffer_window(a.in, (size_t)IN_BPS * (RUNS ? fcur : u * FPW) * a.hop, u < n_units ? total_in : 0), in_voff, raw);
        balance_vmcnt();

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
            if constexpr (WIN_DC && !DC_IN_REGS) {
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
        // multiplies, one rounding more on the composite ones.
        cf twl[Cfg::TWR ? (RL - 1) * CL : 1];
        if constexpr (Cfg::TWR) {
            const cf *hi = lds_all + Cfg::LDS_HI, *lo = lds_all + Cfg::LDS_LO;
            auto root = [&](unsigned m) -> cf { return pk_cmul(hi[m >> 6], lo[m & 63u]); };  // W^m, m < N * RL
            constexpr int RB = RL >= 16 ? 8 : (RL >= 4 ? 2 : RL);  // low digit of r
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
                ebase[c] = (WIN == 0 && (n0 & 1u)) ? -e : e;  // windowed: the (-1)^n rides on the weights (FftArgs::win)
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
            balance_vmcnt();
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
                        if constexpr (WIN != 0) {
                            // a taper whose DC spectrum does not fit the win_dc band: the shifter's + 0.5 (1 + i) -- 128 in
                            // byte units -- goes through the transform with the sample (the u8 kernels' offset-binary form)
                            if (a.win_offset != 0) v[r * C0 + c] += cf{128.0f, 128.0f};
                        }
                    }
                }
            } else if constexpr (WIN != 0 && IN == IN_F32) {
#pragma unroll
                for (int r = 0; r < R0; ++r) convert_row<IN, C0, false>(raw[r], 0u, C0 * t, v + r * C0);  // sign: in the weights
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
                for (int c = 0; c < C0; ++c) dft_regs<R0, C0, (Cfg::ABL & abl::NO_FLOPS) != 0>(v + c);
            }
            lds_write<0>(lds, v, t);
            frame_sync();
            if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0 && iter == 0) a.trace[32 * b + 7] = wall_clock64();  // first pass 0 done
            // prefetch: the next unit is known to every lane now; its bytes stay in flight
            // during the rest of the transform
            size_t un = u + gridDim.x;
#if FSEA_STATIC_CONTIG
            if constexpr (!RUNS) un = u + 1 < contig_end ? u + 1 : n_units;
#endif
            [[maybe_unused]] size_t fnext = 0;
            [[maybe_unused]] bool same_run = false;
            if constexpr (RUNS) {
                same_run = (jpos + 1 < a.run_len) && (fcur + 1 < a.n_frames);
                un = same_run ? u : u + gridDim.x;
                fnext = same_run ? fcur + 1 : un * (size_t)a.run_len;
            }
            if (dyn) {
                if constexpr (Cfg::ABL & abl::NO_LDS) __syncthreads();  // the ablation removed the barrier that publishes tk
                const unsigned nu = __builtin_amdgcn_readfirstlane(tk[par]);
                par ^= 1u;
                un = (nu == NO_UNIT) ? n_units : (size_t)nu;
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
            } else if constexpr ((Cfg::ABL & abl::NO_LOADS) == 0 && !LATE_LOAD) {  // (NO_LOADS, measurement only: the first unit's bytes are reused)
                load_raw(buffer_window(a.in, (size_t)IN_BPS * (un * FPW) * a.hop, un < n_units ? total_in : 0), in_voff, raw);
            }
            middle_pass<1>(lds, lds_all, v, a, t, tw1);

            // last pass
            lds_read<LAST>(lds, v, tl);
            after_reads();
            frame_sync();  // the buffer is free for the next frame's pass 0
            if constexpr (Cfg::TWR && TW_FUSE) {
#pragma unroll
                for (int c = 0; c < CL; ++c) dft_regs_tw<RL, CL, CL>(v + c, twl + c, PRESCALED ? SC : 1.0f);
            } else if constexpr (Cfg::TWR) {
                if constexpr (PRESCALED) {
#pragma unroll
                    for (int c = 0; c < CL; ++c) v[c] = v[c] * cf{SC, SC};  // row 0 has no twiddle
                }
#pragma unroll
                for (int r = 1; r < RL; ++r) {
#pragma unroll
                    for (int c = 0; c < CL; ++c) {
                        if constexpr (Cfg::ABL & abl::NO_FLOPS) v[r * CL + c] += twl[(r - 1) * CL + c];
                        else v[r * CL + c] = pk_cmul(v[r * CL + c], twl[(r - 1) * CL + c]);
                    }
                }
            } else {
                apply_twiddles<LAST>(v, a.tw[LAST], tl);
            }
            if constexpr (!(Cfg::TWR && TW_FUSE)) {
#pragma unroll
                for (int c = 0; c < CL; ++c) dft_regs<RL, CL, (Cfg::ABL & abl::NO_FLOPS) != 0>(v + c);
            }
            if constexpr (DC_IN_REGS) {
#pragma unroll
                for (int c = 0; c < CL; ++c) {
                    v[(RL / 2 - 1) * CL + c] += dcr[c];
                    v[(RL / 2) * CL + c] += dcr[CL + c];
                }
            }
            if constexpr (WIN_DC && !DC_IN_REGS) {
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
            }
            if constexpr (WIN != 0) {
                if constexpr (WIN == 1) {
                    // the next frame's weights: requested in front of this frame's row stores (loads return in order, and a
                    // wait for a load issued behind the stores would wait for those as well)
                    load_window(win_rs, t, wv);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            epilogue(mode, buffer_window(a.out, (size_t)esz * row_elem(RUNS ? fcur : u * FPW), total_out), out_elem, v, tl);
            if constexpr (LATE_LOAD) {
                load_raw(buffer_window(a.in, (size_t)IN_BPS * (un * FPW) * a.hop, un < n_units ? total_in : 0), in_voff, raw);
            }
            u = un;
            if constexpr (RUNS) {
                fcur = fnext;
                jpos = same_run ? jpos + 1 : 0;
            }
            if ((FSEA_TRACE != 0) && a.trace != nullptr && tid == 0 && iter < 24) a.trace[32 * b + 8 + iter] = wall_clock64();
            ++iter;
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
