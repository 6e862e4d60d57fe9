// TEST-ONLY stand-in for frequensea_amd/csrc/fsea_pk_asm_tune.h (the W64 schedule's cross-lane primitives).
#pragma once

namespace fsea {

// cross-lane primitives of the single-wave 64 x 64 schedule: every lane of the emulated wavefront publishes its
// value, then reads the lane it needs (emu_main.cpp)
unsigned emu_lane_read(unsigned v, int src_lane);  // src_lane: lane index inside this work-item's wavefront
static inline void lane_swap32(uint32_t &a, uint32_t &b) {
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t pa = emu_lane_read(a, lane ^ 32), pb = emu_lane_read(b, lane ^ 32);
    if (lane >= 32) a = pb;   // a[32..63] <- b[0..31]
    else b = pa;              // b[0..31]  <- a[32..63]
}
static inline uint32_t quad_xor1(uint32_t v) { return emu_lane_read(v, (int)((threadIdx.x & 63) ^ 1)); }
static inline uint32_t quad_xor2(uint32_t v) { return emu_lane_read(v, (int)((threadIdx.x & 63) ^ 2)); }
static inline uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
    const uint64_t both = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) r |= (uint32_t)((both >> (8 * ((sel >> (8 * i)) & 7))) & 0xff) << (8 * i);
    return r;
}
static inline uint32_t read_lane(uint32_t v, int lane) { return emu_lane_read(v, lane); }

}  // namespace fsea
