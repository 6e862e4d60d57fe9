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


// v_cvt_pk_u8_f32: f converted to u8 (saturating; the argument is already integral) into byte `pos` of `old`
__device__ __forceinline__ uint32_t cvt_pk_u8(float f, uint32_t pos, uint32_t old) {
    return __builtin_amdgcn_cvt_pk_u8_f32(f, pos, old);
}
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
