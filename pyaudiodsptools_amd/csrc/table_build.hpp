// table_build.hpp - host-side construction of the kernels' tables (float64 arithmetic, stored as float or double): the pass
// twiddles of a plan and the spectrum-stage "pair" tables of a filter spectrum.  Shared by the streaming engines (adsp_capi.hip)
// and the uniformly partitioned engines (adsp_upols.hip).  Host code only.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "plan_table.hpp"

namespace adsp {
namespace tables {

// forward-sign twiddles for passes 1.. of the forward then the inverse radix order
// float4 entries (w_{2h+1}, w_{2h+2}) indexed [h * S + jlo], h < R/2; the last one's second half is unused
// (R = float, or double for the f64 flavour of the kernels: ADSP_FORMAT_S16_F64 engines)
template <class R>
struct Vec;
template <>
struct Vec<float> {
    using T2 = float2;
    using T4 = float4;
    static T2 m2(double a, double b) { return make_float2((float)a, (float)b); }
    static T4 m4(double a, double b, double c, double d) { return make_float4((float)a, (float)b, (float)c, (float)d); }
};
template <>
struct Vec<double> {
    using T2 = double2;
    using T4 = double4;
    static T2 m2(double a, double b) { return make_double2(a, b); }
    static T4 m4(double a, double b, double c, double d) { return make_double4(a, b, c, d); }
};

template <class RT>
void build_twiddles(const PlanInfo& pl, std::vector<typename Vec<RT>::T4>& tw) {
    using V = Vec<RT>;
    using T2 = typename V::T2;
    tw.clear();
    auto tw1 = [](int q, int jlo, int R, int S) {
        const double ang = -2.0 * M_PI * (double)q * (double)jlo / ((double)R * (double)S);
        return V::m2(std::cos(ang), std::sin(ang));
    };
    for (int dir = 0; dir < 2; ++dir) {
        int S = 1;
        for (int p = 0; p < pl.NP; ++p) {
            const int R = dir == 0 ? pl.rad[p] : pl.rad[pl.NP - 1 - p];
            if (p > 0 && R == 16 && S >= ADSP_TW2_MIN_S) {
                // two-level: (w^1,w^2), (w^3,w^4), (w^8,w^12) per jlo; the kernel forms w^(4a+b) = w^(4a) w^b
                const int qs[3][2] = {{1, 2}, {3, 4}, {8, 12}};
                for (int h = 0; h < 3; ++h)
                    for (int jlo = 0; jlo < S; ++jlo) {
                        const T2 a = tw1(qs[h][0], jlo, R, S), b = tw1(qs[h][1], jlo, R, S);
                        tw.push_back(V::m4(a.x, a.y, b.x, b.y));
                    }
            } else if (p > 0 && R == 32 && ADSP_TW2_RADIX32 && S >= ADSP_TW2_MIN_S) {
                // two-level, radix 32: w^1..w^8, w^16, w^24 per jlo; the kernel forms w^(8a+b) = w^(8a) w^b
                const int qs[5][2] = {{1, 2}, {3, 4}, {5, 6}, {7, 8}, {16, 24}};
                for (int h = 0; h < 5; ++h)
                    for (int jlo = 0; jlo < S; ++jlo) {
                        const T2 a = tw1(qs[h][0], jlo, R, S), b = tw1(qs[h][1], jlo, R, S);
                        tw.push_back(V::m4(a.x, a.y, b.x, b.y));
                    }
            } else if (p > 0) {
                for (int h = 0; h < R / 2; ++h)
                    for (int jlo = 0; jlo < S; ++jlo) {
                        const T2 a = tw1(2 * h + 1, jlo, R, S);
                        const T2 b = (2 * h + 2 < R) ? tw1(2 * h + 2, jlo, R, S) : V::m2(1.0, 0.0);
                        tw.push_back(V::m4(a.x, a.y, b.x, b.y));
                    }
            }
            S *= R;
        }
    }
}

template <class R>
struct PairEntry {
    typename Vec<R>::T2 wc, g1, g2;
};

template <class R, class HT>
PairEntry<R> pair_entry(const HT* H, int M, int k) {
    // the three entries (c1, c2, c4) of the 2x2 matrix of fftconv_kernel.hpp::pair_op, stored as (wc, g1, g2):
    //   wc' = -i exp(-i pi k/M), g1 = H[k]/4M, g2 = conj(H[M-k])/4M, s = g1+g2, d = g1-g2
    //   c1 = 2s + 2d Re(wc'), c2 = -2i d Im(wc'), c4 = 2s - 2d Re(wc')
    const double ang = M_PI * (double)k / (double)M;
    const double sc = 1.0 / (4.0 * (double)M);
    const double wr = -std::sin(ang), wi = -std::cos(ang);
    const double g1r = (double)H[2 * k] * sc, g1i = (double)H[2 * k + 1] * sc;
    const double g2r = (double)H[2 * (M - k)] * sc, g2i = -(double)H[2 * (M - k) + 1] * sc;
    const double sr = g1r + g2r, si = g1i + g2i, dr = g1r - g2r, di = g1i - g2i;
    PairEntry<R> e;
    e.wc = Vec<R>::m2(2 * sr + 2 * dr * wr, 2 * si + 2 * di * wr);  // c1
    e.g1 = Vec<R>::m2(2 * di * wi, -2 * dr * wi);                    // c2 = -2i d Im(wc')
    e.g2 = Vec<R>::m2(2 * sr - 2 * dr * wr, 2 * si - 2 * di * wr);  // c4
    return e;
}


// The spectrum-stage tables of one plan: `tab` (per-thread rows) and `tab0` (the self-paired butterflies), from the spectrum H of
// M + 1 bins.  Shared by the engine's own plan and by the plan a live session runs on (adsp_live_start).
template <class R, class HT>
void build_pair_tables(const PlanInfo& pl, int M, const HT* H, bool real_spec, std::vector<typename Vec<R>::T4>& tab,
                       std::vector<typename Vec<R>::T2>& tab0) {
    using V = Vec<R>;
    const int T = pl.T;
    const int RR = pl.rad[pl.NP - 1];        // radix of the paired passes: P for XL plans, P/2, P/4 .. otherwise
    const int D = M / RR;                     // bin spacing between a butterfly's outputs
    const int PU = pl.XL ? 1 : pl.P / RR / 2; // pairs of butterflies per thread (in-register plans: (u*T + t, its mirror))
    const int npairs = pl.XL ? RR / 2 : RR;    // pair ops per regular thread and pair of butterflies
    auto first_bin = [&](int t, int u) {     // the butterfly whose outputs thread t pairs (k = bin + D*r)
        if (!pl.XL) return u * T + t;
        const int lo = 32 * (t >> 6) + (t & 31);
        return (t & 32) ? (t == 32 ? T / 2 : T - lo) : lo;
    };
    auto self_paired = [&](int t, int u) { return (t == 0 && u == 0) || (pl.XL && t == 32); };  // served by tab0
    // float4 layout [u][h][3][T]: (wc,g1) of pair 2h, (g2 of 2h, wc of 2h+1), (g1,g2) of 2h+1
    tab.assign((size_t)PU * (npairs / 2) * 3 * T, V::m4(0, 0, 0, 0));
    if (real_spec) {
        // [u][g][3][T] float4 = (c1.re, c4.re, c2.im) of pairs 4g .. 4g+3
        std::vector<double> flat(12);
        for (int u = 0; u < PU; ++u)
            for (int g = 0; g < npairs / 4; ++g)
                for (int tid = 0; tid < T; ++tid) {
                    // XL plans with 8 points per thread run the two self-paired butterflies through the regular pair operations
                    // (spectrum_stage_xl): lane 32 (bins D/2 + D r, partner 7-r in the same lane) takes rows like any other lane;
                    // lane 0 (bins D r) too, except pair 0 = the bins 0 and M/2, each its own partner: (c1(0), c1(M/2), Im c2(0))
                    const bool in_lane = pl.XL && RR == 8 && self_paired(tid, u);
                    if (self_paired(tid, u) && !in_lane) continue;
                    for (int q = 0; q < 4; ++q) {
                        const PairEntry<R> pe = pair_entry<R>(H, M, first_bin(tid, u) + D * (4 * g + q));
                        flat[3 * q + 0] = pe.wc.x;
                        flat[3 * q + 1] = pe.g2.x;
                        flat[3 * q + 2] = pe.g1.y;
                    }
                    if (in_lane && tid == 0) flat[1] = pair_entry<R>(H, M, M / 2).wc.x;
                    for (int j = 0; j < 3; ++j)
                        tab[((size_t)(u * (npairs / 4) + g) * 3 + j) * T + tid] =
                            V::m4(flat[4 * j], flat[4 * j + 1], flat[4 * j + 2], flat[4 * j + 3]);
                }
    }
    for (int u = 0; u < PU && !real_spec; ++u)
        for (int h = 0; h < npairs / 2; ++h)
            for (int tid = 0; tid < T; ++tid) {
                const bool in_lane = pl.XL && RR == 8 && self_paired(tid, u);  // (as above)
                if (self_paired(tid, u) && !in_lane) continue;                  // self-paired butterflies: tab0
                PairEntry<R> a = pair_entry<R>(H, M, first_bin(tid, u) + D * (2 * h));
                const PairEntry<R> b = pair_entry<R>(H, M, first_bin(tid, u) + D * (2 * h + 1));
                if (in_lane && tid == 0 && h == 0) {  // bins 0 and M/2: c4 <- conj(c1(M/2)), which then multiplies register 4
                    const PairEntry<R> m = pair_entry<R>(H, M, M / 2);
                    a.g2 = V::m2(m.wc.x, -m.wc.y);
                }
                const size_t row = (size_t)(u * (npairs / 2) + h) * 3;
                tab[(row + 0) * T + tid] = V::m4(a.wc.x, a.wc.y, a.g1.x, a.g1.y);
                tab[(row + 1) * T + tid] = V::m4(a.g2.x, a.g2.y, b.wc.x, b.wc.y);
                tab[(row + 2) * T + tid] = V::m4(b.g1.x, b.g1.y, b.g2.x, b.g2.y);
            }
    tab0.assign((size_t)(RR + 1) * 3, V::m2(0, 0));
    auto put0 = [&](int idx, int k) {
        const PairEntry<R> pe = pair_entry<R>(H, M, k);
        tab0[idx * 3 + 0] = pe.wc;
        tab0[idx * 3 + 1] = pe.g1;
        tab0[idx * 3 + 2] = pe.g2;
    };
    put0(0, 0);
    put0(1, M / 2);
    for (int r = 1; r < RR / 2; ++r) put0(2 + (r - 1), D * r);
    for (int r = 0; r < RR / 2; ++r) put0(2 + (RR / 2 - 1) + r, D / 2 + D * r);
}


}  // namespace tables
}  // namespace adsp
