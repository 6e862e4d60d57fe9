// fsea_pk_asm.h -- the packed complex type and the one operation the compiler
// cannot select optimally from C++: a complex multiply by a run-time twiddle.
// hipcc folds swizzles and whole-vector negations into op_sel / neg modifiers
// of v_pk_* instructions, but not the single-lane negation a complex product
// needs, so that product is written as two VOP3P instructions by hand:
//   t   = a * (w.x, w.x)                              v_pk_mul_f32, op_sel broadcast of w.x
//   t.x = -a.y * w.y + t.x ; t.y = a.x * w.y + t.y    v_pk_fma_f32, swapped a, broadcast w.y,
//                                                     neg_lo on the w operand
// Register-only, non-volatile asm: the scheduler may move it freely; plain VALU
// RAW dependencies are interlocked in hardware, no wait states are needed.
#pragma once

#include <stdint.h>

namespace fsea {

typedef float cf __attribute__((vector_size(8)));    // (re, im) in an aligned VGPR pair
typedef float cf2 __attribute__((vector_size(16)));  // two of them, for 16-byte memory ops

__device__ __forceinline__ cf pk_cmul(cf a, cf w) {
    cf t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(t) : "v"(a), "v"(w));
    return t;
}

// a * w with a wavefront-uniform w held in an SGPR pair (one constant-bus operand per instruction).
__device__ __forceinline__ cf pk_cmul_uniform(cf a, cf w) {
    cf t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(t) : "v"(a), "s"(w));
    return t;
}

// a complex value times a real weight: a * wp.x (pk_scale_lo) or a * wp.y (pk_scale_hi), the weight broadcast to both
// halves by op_sel -- the taper window's multiply: weights stay packed two to a register pair, as they are loaded.
__device__ __forceinline__ cf pk_scale_lo(cf a, cf wp) {
    cf t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(wp));
    return t;
}
__device__ __forceinline__ cf pk_scale_hi(cf a, cf wp) {
    cf t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(t) : "v"(a), "v"(wp));
    return t;
}

// a * (weight) + c and a * (weight) - c with the real weight broadcast from the low / high half of wp: the first butterfly
// level of pass 0 with the taper folded in, (wa a) +- (wb b) as one packed multiply and two packed FMAs for the pair.
__device__ __forceinline__ cf pk_wfma_lo(cf a, cf wp, cf c) {
    cf t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(t) : "v"(a), "v"(wp), "v"(c));
    return t;
}
__device__ __forceinline__ cf pk_wfma_hi(cf a, cf wp, cf c) {
    cf t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(t) : "v"(a), "v"(wp), "v"(c));
    return t;
}
__device__ __forceinline__ cf pk_wfms_lo(cf a, cf wp, cf c) {
    cf t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(t) : "v"(a), "v"(wp), "v"(c));
    return t;
}
__device__ __forceinline__ cf pk_wfms_hi(cf a, cf wp, cf c) {
    cf t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(t) : "v"(a), "v"(wp), "v"(c));
    return t;
}

// c + a * w with the same two-instruction shape (the product's first half takes c as addend).
__device__ __forceinline__ cf pk_cmul_add(cf a, cf w, cf c) {
    cf t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(t) : "v"(a), "v"(w), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(t) : "v"(a), "v"(w));
    return t;
}

// a + (-i) b = (a.x + b.y, a.y - b.x) and a + i b = (a.x - b.y, a.y + b.x): the +-i butterflies as plain
// packed adds (swizzle and one-lane negation in the modifiers; no multiplier involved).
__device__ __forceinline__ cf pk_add_mi(cf a, cf b) {
    cf t;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    return t;
}
__device__ __forceinline__ cf pk_add_pi(cf a, cf b) {
    cf t;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    return t;
}

// c + a * (-i w): the same product with the twiddle rotated by -i = W^{L/4}, (-i w) = (w.y, -w.x),
// selected by modifiers alone, so that the deferred-twiddle butterflies (dft_regs_def) keep one
// register pair per twiddle pair {w, -i w}.
__device__ __forceinline__ cf pk_cmul_add_mi(cf a, cf w, cf c) {
    cf t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(t) : "v"(a), "v"(w), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]" : "+v"(t) : "v"(a), "v"(w));
    return t;
}


// ---- the last butterfly level in "power form" (FftCfg OPT 8388608) ----
// For the butterfly pair x0 = a + w b, x1 = a - w b of the last level of the last pass, whose outputs only ever enter
// |x|^2, the two results are formed planar instead of as two complex values: re = (x0.re, x1.re), im = (x0.im, x1.im),
// so that |x0|^2 and |x1|^2 are ONE v_pk_mul_f32 + ONE v_pk_fma_f32 for the pair instead of v_mul + v_fmac per bin.
// The broadcasts and the one-lane negations are op_sel / neg modifiers; operation order per component is the one of
// bfly_const (x0 bit-identical; x1 = a - w b directly instead of 2 a - x0).
// w = 1
__device__ __forceinline__ cf pk_pm_re(cf a, cf b) {  // (a.x + b.x, a.x - b.x)
    cf t;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    return t;
}
__device__ __forceinline__ cf pk_pm_im(cf a, cf b) {  // (a.y + b.y, a.y - b.y)
    cf t;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    return t;
}
// w = -i: w b = (b.y, -b.x)
__device__ __forceinline__ cf pk_pm_re_mi(cf a, cf b) {  // (a.x + b.y, a.x - b.y)
    cf t;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    return t;
}
__device__ __forceinline__ cf pk_pm_im_mi(cf a, cf b) {  // (a.y - b.x, a.y + b.x)
    cf t;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    return t;
}
// constant w = (w.x, w.y), wavefront-uniform in an SGPR pair: w b = (w.x b.x - w.y b.y, w.x b.y + w.y b.x)
__device__ __forceinline__ cf pk_pm_re_w(cf a, cf b, cf w) {  // (a.x + (w b).x, a.x - (w b).x)
    cf t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,0,0] neg_hi:[0,1,0]" : "=v"(t) : "v"(b), "s"(w), "v"(a));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0]" : "+v"(t) : "v"(b), "s"(w));
    return t;
}
__device__ __forceinline__ cf pk_pm_im_w(cf a, cf b, cf w) {  // (a.y + (w b).y, a.y - (w b).y)
    cf t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "=v"(t) : "v"(b), "s"(w), "v"(a));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "+v"(t) : "v"(b), "s"(w));
    return t;
}


// ---- cross-lane and byte primitives of the single-wave 64 x 64 schedule (FftKernel::run_w64) ----
// v_permlane32_swap_b32: lanes 32-63 of `a` trade places with lanes 0-31 of `b` (a half exchange; the other two
// halves stay).  The builtin lets hipcc place the wait states its operands need behind a VALU write.
__device__ __forceinline__ void lane_swap32(uint32_t &a, uint32_t &b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
// the value of lane ^ 1 / lane ^ 2 inside each quad (v_mov_b32_dpp quad_perm:[1,0,3,2] / [2,3,0,1])
__device__ __forceinline__ uint32_t quad_xor1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t quad_xor2(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
}
// v_perm_b32: result byte i = byte sel[8i+7:8i] of the eight bytes {hi, lo} (0-3 = lo, 4-7 = hi)
__device__ __forceinline__ uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
    return __builtin_amdgcn_perm(hi, lo, sel);
}
// v_cvt_pk_u8_f32: f converted to u8 (saturating; the argument is already integral) into byte `pos` of `old`
__device__ __forceinline__ uint32_t cvt_pk_u8(float f, uint32_t pos, uint32_t old) {
    return __builtin_amdgcn_cvt_pk_u8_f32(f, pos, old);
}
__device__ __forceinline__ uint32_t read_lane(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ float trunc_f32(float x) { return __builtin_truncf(x); }

// Diagnostics only (FSEA_TRACE): where a workgroup runs.
__device__ __forceinline__ unsigned read_hw_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}
__device__ __forceinline__ unsigned read_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v;
}

}  // namespace fsea
