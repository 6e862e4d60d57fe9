// fsea_pk_asm_tune.h -- cross-lane and byte primitives that only the tuning library's schedules use (opt::W64:
// FftKernel::run_w64 / pixels_w64, fsea_fft_tune_members.h).  Included by fsea_fft_core.h under -DFSEA_TUNE only.
#pragma once

#include <stdint.h>

namespace fsea {

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
__device__ __forceinline__ uint32_t read_lane(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }

}  // namespace fsea
