// fsea_tables.h -- host-side twiddle tables for the Stockham passes.
// Pass i >= 1 uses W^{r k} with W = exp(-2 pi i / (Ns_i R_i)), stored as
// [(r-1) * Ns_i + k], r = 1..R_i-1, k = 0..Ns_i-1; passes are concatenated in
// order (pass 1 first), which is also the order the kernel copies the middle
// passes into LDS.  Angles are reduced exactly in integers and evaluated in
// double before rounding to f32.
#pragma once

#include <cmath>
#include <cstddef>
#include <vector>

namespace fsea {

struct TwPair {
    float re, im;
};

inline void build_twiddles(int np, const int *radix, std::vector<TwPair> &tw, size_t *offsets) {
    const double two_pi = 6.283185307179586476925286766559;
    long long ns = radix[0];
    tw.clear();
    offsets[0] = 0;
    for (int i = 1; i < np; ++i) {
        offsets[i] = tw.size();
        const long long r_i = radix[i];
        const long long len = ns * r_i;
        for (long long r = 1; r < r_i; ++r) {
            for (long long k = 0; k < ns; ++k) {
                const double ang = -two_pi * (double)((r * k) % len) / (double)len;
                tw.push_back(TwPair{(float)std::cos(ang), (float)std::sin(ang)});
            }
        }
        ns *= r_i;
    }
    for (int i = np; i < 4; ++i) offsets[i] = tw.size();
}

}  // namespace fsea
