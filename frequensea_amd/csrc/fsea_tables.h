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

// Taper window tables of the windowed kernels (FftArgs::win, win_dc; fsea_plan_set_window).  n points over t lanes, first
// radix r0, last radix rl; w: n weights.
//   perm: (-1)^j w[j] in the order the pass-0 lanes hold their samples: register i = r c0 + c of lane t is sample
//         j = c0 t + c + r n/r0 and sits at perm[(i / 4) 4 t_lanes + 4 t + i % 4].
//   dc:   the spectrum of the offset-binary DC term, S[k] = 0.5 (1 + i) D[k] with D the DFT of (-1)^j w[j] (double,
//         iterative radix 2), for the 2 nsl bins [n/2 - nsl, n/2 + nsl), nsl = n / rl: the rows rl/2 - 1 and rl/2 of the
//         last pass.  Entries below 2^-40 of the peak are zero, so that a rectangular w gives the un-windowed kernels' bits.
// Returns the form: 1 (centred: the kernels transform w (u8 - 128) and add dc) when S is confined to that band, else 2
// (offset-binary: w u8, dc all zero).  "Confined": the part of S the centred form leaves out -- the bins outside the band
// -- has an rms of at most 2^-24 sqrt(0.5 sum w^2), half of what the offset-binary form's f32 rounding puts into every bin
// (relative L2 error of the transform ~1.2e-7, of a spectrum whose energy is the DC term's, n * 0.5 sum w^2).  Cosine-sum
// tapers rounded to f32 leave ~2e-8 there (the rounding of the weights itself, not confined to any band); Kaiser and
// truncated Gaussians leave their skirts (1e-6 ... 1e-4) and take form 2.
inline int build_window_tables(int n, int t_lanes, int r0, int rl, const float *w, std::vector<float> &perm,
                               std::vector<TwPair> &dc) {
    const int p = n / t_lanes, c0 = p / r0;
    perm.assign((size_t)n, 0.0f);
    for (int t = 0; t < t_lanes; ++t) {
        for (int r = 0; r < r0; ++r) {
            for (int c = 0; c < c0; ++c) {
                const int j = c0 * t + c + r * (n / r0);
                const int i = r * c0 + c;
                perm[(size_t)(i / 4) * 4 * (size_t)t_lanes + (size_t)4 * t + (size_t)(i % 4)] = (j & 1) ? -w[j] : w[j];
            }
        }
    }
    std::vector<double> re((size_t)n), im((size_t)n, 0.0);
    int bits = 0;
    while ((1 << bits) < n) ++bits;
    for (int j = 0; j < n; ++j) {
        int rj = 0;
        for (int b = 0; b < bits; ++b) rj |= ((j >> b) & 1) << (bits - 1 - b);
        re[(size_t)rj] = (j & 1) ? -(double)w[j] : (double)w[j];
    }
    const double two_pi = 6.283185307179586476925286766559;
    for (int len = 2; len <= n; len <<= 1) {
        const int half = len / 2;
        for (int k = 0; k < half; ++k) {
            const double wr = std::cos(two_pi * (double)k / (double)len), wi = -std::sin(two_pi * (double)k / (double)len);
            for (int base = k; base < n; base += len) {
                const size_t i0 = (size_t)base, i1 = (size_t)base + (size_t)half;
                const double tr = re[i1] * wr - im[i1] * wi, ti = re[i1] * wi + im[i1] * wr;
                re[i1] = re[i0] - tr;
                im[i1] = im[i0] - ti;
                re[i0] += tr;
                im[i0] += ti;
            }
        }
    }
    const int nsl = n / rl;
    double peak = 0.0, outside2 = 0.0, w2 = 0.0;
    int n_outside = 0;
    for (int k = 0; k < n; ++k) {
        const double m = std::hypot(0.5 * (re[(size_t)k] - im[(size_t)k]), 0.5 * (re[(size_t)k] + im[(size_t)k]));
        if (m > peak) peak = m;
        if (k < n / 2 - nsl || k >= n / 2 + nsl) {
            outside2 += m * m;
            ++n_outside;
        }
        w2 += (double)w[k] * (double)w[k];
    }
    const double rms_outside = n_outside ? std::sqrt(outside2 / n_outside) : 0.0;
    const int form = (rms_outside <= std::ldexp(std::sqrt(0.5 * w2), -24)) ? 1 : 2;
    dc.assign((size_t)2 * (size_t)nsl, TwPair{0.f, 0.f});
    if (form == 1) {
        for (int j = 0; j < 2 * nsl; ++j) {
            const size_t k = (size_t)(n / 2 - nsl + j);
            const double sr = 0.5 * (re[k] - im[k]), si = 0.5 * (re[k] + im[k]);
            if (std::hypot(sr, si) > std::ldexp(peak, -40)) dc[(size_t)j] = TwPair{(float)sr, (float)si};
        }
    }
    return form;
}

}  // namespace fsea
