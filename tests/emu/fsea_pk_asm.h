// TEST-ONLY stand-in for frequensea_amd/csrc/fsea_pk_asm.h (found first on the emulation
// build's include path): same types, the complex multiply written in plain C++.
#pragma once

namespace fsea {

typedef float cf __attribute__((vector_size(8)));
typedef float cf2 __attribute__((vector_size(16)));

static inline cf pk_cmul(cf a, cf w) {
    cf t = cf{a[0] * w[0], a[1] * w[0]};
    t = cf{fmaf(-a[1], w[1], t[0]), fmaf(a[0], w[1], t[1])};
    return t;
}

static inline cf pk_cmul_uniform(cf a, cf w) { return pk_cmul(a, w); }
static inline cf pk_scale_lo(cf a, cf wp) { return cf{a[0] * wp[0], a[1] * wp[0]}; }
static inline cf pk_scale_hi(cf a, cf wp) { return cf{a[0] * wp[1], a[1] * wp[1]}; }
static inline cf pk_wfma_lo(cf a, cf wp, cf c) { return cf{fmaf(a[0], wp[0], c[0]), fmaf(a[1], wp[0], c[1])}; }
static inline cf pk_wfma_hi(cf a, cf wp, cf c) { return cf{fmaf(a[0], wp[1], c[0]), fmaf(a[1], wp[1], c[1])}; }
static inline cf pk_wfms_lo(cf a, cf wp, cf c) { return cf{fmaf(a[0], wp[0], -c[0]), fmaf(a[1], wp[0], -c[1])}; }
static inline cf pk_wfms_hi(cf a, cf wp, cf c) { return cf{fmaf(a[0], wp[1], -c[0]), fmaf(a[1], wp[1], -c[1])}; }

static inline cf pk_cmul_add(cf a, cf w, cf c) {
    cf t = cf{fmaf(a[0], w[0], c[0]), fmaf(a[1], w[0], c[1])};
    t = cf{fmaf(-a[1], w[1], t[0]), fmaf(a[0], w[1], t[1])};
    return t;
}

static inline cf pk_add_mi(cf a, cf b) { return cf{a[0] + b[1], a[1] - b[0]}; }
static inline cf pk_add_pi(cf a, cf b) { return cf{a[0] - b[1], a[1] + b[0]}; }

static inline cf pk_cmul_add_mi(cf a, cf w, cf c) {
    cf t = cf{fmaf(a[0], w[1], c[0]), fmaf(a[1], w[1], c[1])};
    t = cf{fmaf(a[1], w[0], t[0]), fmaf(-a[0], w[0], t[1])};
    return t;
}

static inline uint32_t cvt_pk_u8(float f, uint32_t pos, uint32_t old) {
    const float c = f < 0.0f ? 0.0f : (f > 255.0f ? 255.0f : f);  // (NaN does not occur: p >= 0)
    const uint32_t u = (uint32_t)nearbyintf(c);                   // round to nearest even, as the instruction
    return (old & ~(0xffu << (8 * pos))) | (u << (8 * pos));
}
static inline float trunc_f32(float x) { return truncf(x); }

static inline unsigned read_hw_id() { return 0; }
static inline unsigned read_xcc_id() { return 0; }

}  // namespace fsea
