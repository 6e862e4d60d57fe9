// fsea_tables.h -- host-side twiddle tables for the Stockham passes.
// Pass i >= 1 uses W^{r k} with W = exp(-2 pi i / (Ns_i R_i)), stored as
// [(r-1) * Ns_i + k], r = 1..R_i-1, k = 0..Ns_i-1.  Angles are reduced exactly in
// integers and evaluated in double before rounding to f32.
#pragma once

#include <cmath>
#include <cstddef>
#include <vector>

namespace fsea {

struct TwPair {
    float re, im;
};

// Image layout (complex f32 entries):
//   [ pass 1 | ... | pass LAST-1 | HI | LO | pass LAST ]
// The part before "pass LAST" is the small block a workgroup copies into LDS: the middle-pass
// tables and the two factors of the last pass, W_N^{64 h} (HI, N/64 entries) and W_N^{l}
// (LO, 64 entries), from which the kernel builds its register-resident last-pass twiddles as
// W_N^{m} = HI[m >> 6] * LO[m & 63].  The full last-pass table follows for the configurations
// that read it from memory every frame.  offsets[i] = start of pass i, offsets[4] = start of HI.
inline void build_twiddles(int np, const int *radix, std::vector<TwPair> &tw, size_t *offsets) {
    const double two_pi = 6.283185307179586476925286766559;
    long long n = 1;
    for (int i = 0; i < np; ++i) n *= radix[i];
    tw.clear();
    auto pass_table = [&](int i) {
        long long ns = 1;
        for (int k = 0; k < i; ++k) ns *= radix[k];
        const long long r_i = radix[i], len = ns * r_i;
        offsets[i] = tw.size();
        for (long long r = 1; r < r_i; ++r) {
            for (long long k = 0; k < ns; ++k) {
                const double ang = -two_pi * (double)((r * k) % len) / (double)len;
                tw.push_back(TwPair{(float)std::cos(ang), (float)std::sin(ang)});
            }
        }
    };
    offsets[0] = 0;
    for (int i = 1; i < np - 1; ++i) pass_table(i);
    offsets[4] = tw.size();
    for (long long h = 0; h < (n >= 64 ? n / 64 : 1); ++h) {
        const double ang = -two_pi * (double)(64 * h) / (double)n;
        tw.push_back(TwPair{(float)std::cos(ang), (float)std::sin(ang)});
    }
    for (long long l = 0; l < 64; ++l) {
        const double ang = -two_pi * (double)l / (double)n;
        tw.push_back(TwPair{(float)std::cos(ang), (float)std::sin(ang)});
    }
    pass_table(np - 1);
    for (int i = np; i < 4; ++i) offsets[i] = tw.size();
}

// Deferred-twiddle table of the V2 schedule's middle pass (dft_regs_def): lane group ka (the
// pass-0 output digit, 0..ra-1) scales its pass-1 inputs x_b by om^b, om = W_{ra rb}^{ka}.  Row
// ka holds rb/2 entries: for half = 1, 2, 4, ..: om^{rb/(2 half)} W_{2 half}^j, j < max(1, half/2),
// i.e. exp(-2 pi i (ka + ra j) / (2 ra half)); angles reduced exactly in integers.
inline void build_deferred_table(int ra, int rb, std::vector<TwPair> &out) {
    const double two_pi = 6.283185307179586476925286766559;
    out.clear();
    for (int ka = 0; ka < ra; ++ka) {
        for (int half = 1; half < rb; half <<= 1) {
            const int distinct = half >= 2 ? half / 2 : 1;
            const long long den = 2LL * ra * half;
            for (int j = 0; j < distinct; ++j) {
                const long long num = ((long long)ka + (long long)ra * j) % den;
                const double ang = -two_pi * (double)num / (double)den;
                out.push_back(TwPair{(float)std::cos(ang), (float)std::sin(ang)});
            }
        }
    }
}

// Fused frequency shift (FftArgs::rot_row): the phasor between consecutive pass-0 rows of one
// lane, rows[r] = e^{2 pi i delta r n / r0}, delta in turns per sample; 32 entries (r0 <= 32).
inline void build_rotation_rows(int n, int r0, double delta, TwPair *rows) {
    const double two_pi = 6.283185307179586476925286766559;
    for (int r = 0; r < 32; ++r) {
        double turns = (r < r0) ? delta * (double)r * (double)(n / r0) : 0.0;
        turns -= std::floor(turns);
        rows[r].re = (float)std::cos(two_pi * turns);
        rows[r].im = (float)std::sin(two_pi * turns);
    }
}

}  // namespace fsea
