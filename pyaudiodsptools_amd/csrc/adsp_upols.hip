// adsp_upols.hip - uniformly partitioned overlap-save: streaming FIRs LONGER than one transform (round 5).
//
// The reference's own GPU example runs chunk_size 88200 (Example4.py:5, ModuleTestsGPU.py:35): a low cut of 44 099 taps, an EQ
// composite of 88 197 - more than the largest transform of fftconv_kernel.hpp (32768 real points) can take in one piece.  Round 1-4
// cut such a kernel into slices of <= 14336 taps and ran ONE ENGINE PER SLICE over the same input (PartitionedFirEngine: P forward
// and P inverse transforms per block, P reads of the input, P - 1 read-modify-writes of the output).  Here the kernel is cut into P
// partitions of B taps each (B = 8192: the M = 8192 plan, the fastest per point; or 16384), every input block of B samples is transformed ONCE,
// its spectrum kept in a frequency-domain delay line in HBM, and an output block is
//
//     y_b = irfft( sum_p  X_{b-p} . H_p )[B .. 2B)          X_b = rfft( s[(b-1)B .. (b+1)B) )
//
// - one forward transform, P spectrum multiply-accumulates, ONE inverse transform and one store per B samples.  Two launches per
// call, each over every (channel, block) at once:
//   upols_forward_kernel : window of 2B samples (generic chunk geometry: any chunk size divisible by 4) -> the plan's forward passes
//                          -> the real-FFT SPLIT of the thread's registers (both partners of a bin pair live in one thread) -> the
//                          delay line.  What is stored is the half spectrum of the block's 2B real samples - A = 2 X[k] in the place
//                          of Z[k], B = 2 conj X[M-k] in the place of Z[M-k] (upols_split) - in the REGISTER LAYOUT of the plan
//                          (element e = tid + T m of the last forward pass), 8 M bytes per block, in 16-byte units per lane, in
//                          the order the second launch reads them.
//   upols_mac_kernel     : acc += g_p . S_{b-p} for p = 0 .. P-1, register by register - ONE complex multiply-add per register and
//                          partition against a table in the same layout (H_p[k] / 4M, or its conjugate on the partners' places:
//                          upols_upload_tables) - then the re-packing for the inverse (upols_merge, once per output block), the
//                          inverse passes, the kept half [B, 2B) converted / passed through a fused effect and stored.  The launch
//                          is bound by what the L1 / L2 path delivers (tables + spectra: 16 P bytes per output sample - 8 of table,
//                          8 of spectrum), not by its arithmetic: rounds 5 - 6a multiplied the UNSPLIT Z by pair_op's 2x2 matrix per
//                          pair and partition (22 vector instructions per pair against 8) and were 1 - 5 % slower per call
//                          (profiles/r6f_upols_split_spectra_ab.txt).
// The second launch reads what the first one wrote: the kernel boundary is the only synchronisation (no flags, no scopes).
// Consecutive blocks of a channel share P - 1 spectra; blockIdx -> (channel, block) keeps them on one XCD (their L2).
//
// Time axis.  The stream is out[tau] = y[tau - delay], y = taps (*) s; blocks tile the y axis from absolute index 0, the chunk grid
// is independent of them (chunk 88200 = 10.77 blocks).  A call for chunks k .. k+n-1 (input complete up to (k+n) N - 1) transforms
// the blocks whose windows have just become complete and produces the output blocks that meet tau in [kN, (k+n)N); a block that
// straddles a call boundary is computed whole by the first of the two calls and its second part carried to the next call's output -
// or, where that does not pay, multiplied and inverse-transformed in both (adsp_upols_set_carry; its forward transform once either way).  delay >= B keeps
// every block an output needs inside the input that has arrived (the reference's devices delay by ~3/4 chunk).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/adsp.h"
#include "capi_common.hpp"
#include "plan_table.hpp"
#include "table_build.hpp"

using adsp::fail;

namespace adsp {  // adsp_rccl.hip
int rccl_broadcast(float* const* d_buf, const int* devs, const hipStream_t* streams, int n, size_t count, int root);
int rccl_broadcast_rank(const char* unique_id, int rank, int world, int root, int dev, float* d_buf, size_t count, hipStream_t stream);
}  // namespace adsp

// tuning: stages of the multiply kernel requested ahead, and its workgroups per CU (32 points per thread; the 64-point plan keeps two and two).
// Rounds 5 - 6a: four stages ahead in two workgroups per CU (181 registers); the split form needs 148 registers with two stages ahead, three
// workgroups fit, and that is 2 - 5 % faster at blocks of 8192 (tools/sessions/r6_session30.sh; four stages ahead at three per CU spill: +45 %)
// (The compiler hoists a stage's successor requests above its multiply-adds, so the loads need a register set of their own: 148 registers at two
// stages ahead.  With a scheduling barrier between them two stages fit 122 registers and four fit 140 - and the time does not move, at either block
// size, at 64 ... 1024 channels: session 38.  The launch is bound by the bytes it pulls through L2 and HBM, not by how far ahead it asks.)
#ifndef ADSP_UPOLS_AHEAD
#define ADSP_UPOLS_AHEAD 2
#endif
#ifndef ADSP_UPOLS_AHEAD_512
#define ADSP_UPOLS_AHEAD_512 1
#endif
#ifndef ADSP_UPOLS_MAC_WAVES
#define ADSP_UPOLS_MAC_WAVES 3
#endif
#ifndef ADSP_UPOLS_FWD_WAVES
#define ADSP_UPOLS_FWD_WAVES 3
#endif
#ifndef ADSP_UPOLS_FWD_NT
#define ADSP_UPOLS_FWD_NT 0
#endif
#ifndef ADSP_UPOLS_ABLATE  // tuning builds only (make tuning EXTRA=-DADSP_UPOLS_ABLATE=<mask>): 1 no table loads, 2 no spectrum loads
#define ADSP_UPOLS_ABLATE 0
#endif
#if ADSP_UPOLS_ABLATE != 0 && !defined(ADSP_TUNING_BUILD)
#error "ADSP_UPOLS_ABLATE changes what the kernels compute: tuning builds only (make tuning)"
#endif

namespace adsp {

struct UpolsArgs {
    const void* ring;    // [ring_slots][C][N] input history ring (samples of the engine's format)
    const void* in;      // [n_steps][C][N]
    void* out;           // [n_steps][C][N]
    const void* zeros;   // N zero samples
    const float4* tw;    // pass twiddles of the plan
    const float4* pair;  // [P] tables in the delay line's layout: [R][T] float4 = g of the unit's two registers (upols_upload_tables), pair_stride float4 each
    float2* zline;       // [C][R][PTS/2][T] float4: the delay line of forward-transformed blocks (two registers per unit: upols_forward_kernel)
    int ring_pos, ring_slots, C, N, nh, n_steps;
    float inv_n;
    int P;               // partitions
    int R;               // delay-line slots per channel
    int slot_first;      // slot of the launch's first block
    int p_first;         // (absolute index of the launch's first output block) mod P: where its partition order starts (multiply launch)
    int nblk;            // blocks in this launch
    int rel_first;       // forward: window start of the first block; mac: output time of the first block's first kept sample - relative to
                         // the first new input sample of the call (may be negative)
    int ncg;
    int pair_stride;
    int epi_op;
    float epi_p0, epi_p1, epi_p2;
    // the multiply launch's extra workgroups (blockIdx >= mac_grid, one per channel) keep the input's tail for the next call's windows
    void* ring_w;        // = ring
    int mac_grid;        // workgroups of the multiply-accumulate proper
    int tail;            // samples per channel to keep: min(2B, n_steps N) - no later window reaches further back (multiple of 4)
    int cnt;             // chunks of this call that enter the ring: min(n_steps, nh)
    int epi_phase;       // fused tremolo (EffectTremolo.py:27-47): LFO table index of the call's first output sample ...
    int epi_replay;      // ... or, 1: every chunk replays the table from epi_phase (the reference's buffer quirk, adsp_capi.hip: tremolo_run)
    // The block that straddles the end of the call is computed ONCE: what it holds beyond the call's last sample goes - as float32, before any
    // effect - into carry_w[C][B], and the next call's per-channel workgroups copy it (effect, sample format) to the head of THEIR output
    float2* carry_w;        // [C][B/2]: written by the launch's last block
    const float2* carry_r;  // what the previous call left ...
    int carry_stride;       // B
    int carry_n;            // ... its first carry_n samples are this call's outputs 0 .. carry_n-1 (0: none - the launch starts with the straddling block)
};

namespace {

// cos(pi x), sin(pi x) for 0 <= x < 1 at compile time (Taylor series about 0 of the argument folded into [-pi/2, pi/2]: 1e-16)
__host__ __device__ constexpr double const_sin_small(double y) {
    double term = y, sum = y;
    for (int n = 1; n < 14; ++n) {
        term *= -y * y / ((2 * n) * (2 * n + 1));
        sum += term;
    }
    return sum;
}
__host__ __device__ constexpr double const_cos_small(double y) {
    double term = 1.0, sum = 1.0;
    for (int n = 1; n < 14; ++n) {
        term *= -y * y / ((2 * n - 1) * (2 * n));
        sum += term;
    }
    return sum;
}
constexpr double kPi = 3.14159265358979323846;
__host__ __device__ constexpr double const_sin_pi(double x) { return x <= 0.5 ? const_sin_small(kPi * x) : const_sin_small(kPi * (1.0 - x)); }
__host__ __device__ constexpr double const_cos_pi(double x) { return x <= 0.5 ? const_cos_small(kPi * x) : -const_cos_small(kPi * (1.0 - x)); }

template <int R>
struct HalfTurn {  // (cos, sin)(pi r / R), r = 0 .. R-1, as float literals of the kernel
    float c[R], s[R];
};
template <int R>
constexpr HalfTurn<R> half_turn() {
    HalfTurn<R> t{};
    for (int r = 0; r < R; ++r) {
        t.c[r] = static_cast<float>(const_cos_pi(static_cast<double>(r) / R));
        t.s[r] = static_cast<float>(const_sin_pi(static_cast<double>(r) / R));
    }
    return t;
}

template <class PL>
__device__ __forceinline__ void upols_indices(int tid, int& ja, int& jb) {
    static_assert(!PL::XL && PL::NBL == 2, "partitioned engines run the in-register pairing plans with two paired butterflies per thread");
    ja = tid;
    jb = (tid == 0) ? PL::NBL * PL::T / 2 : PL::NBL * PL::T - tid;
}

template <class PL>
__device__ __forceinline__ bool upols_block(const UpolsArgs& a, int& c, int& blk) {
    const int lin = static_cast<int>(blockIdx.x);
    const int xcd = lin & 7;
    const int idx = lin >> 3;
    const int cgl = idx / a.nblk;
    blk = idx - cgl * a.nblk;
    c = cgl * 8 + xcd;  // one channel per workgroup; a channel's consecutive blocks stay on one XCD
    return c < a.C;
}


// The real-FFT split and its inverse, on the register layout of the in-register pairing plans (round 6, second half).  Register NB r of
// thread t > 0 holds Z[k], k = t + (M/R) r, and register NB (R-1-r) + 1 its partner Z[M-k].  With U = Za + conj Zb, D = Za - conj Zb and
// wc = -i exp(-i pi k / M) (fftconv_core.inc: pair_op)
//     A = U + wc D  (= 2 X[k]),   B = U - wc D  (= 2 conj X[M-k])
// are what the delay line keeps in the two registers' places (upols_split, once per input block), the multiply launch accumulates
//     SP = sum_p g1_p A_{b-p},  SQ = sum_p g2_p B_{b-p}          g1 = H_p[k] / 4M,  g2 = conj(H_p[M-k]) / 4M
// - ONE complex multiply-add per register and partition against a table that has the registers' layout - and
//     Zy[k] = E + O,  Zy[M-k] = conj(E - O),   E = SP + SQ,  O = conj(wc) (SP - SQ)
// (upols_merge, once per output block) is what the inverse passes take: pair_op's matrix c1 = 2s + 2d Re wc, c2 = -2i d Im wc, c4 = 2s - 2d Re wc
// factored into its three steps, of which only the middle one depends on the partition.  (Rounds 5 - 6a summed pair_op_p(Z_{b-p}) on the
// unsplit Z: 22 vector instructions per pair and partition - 16 for the matrix, 6 to form it from (2s, 2d) and the twiddle - against 8 here.)
// Thread 0's two butterflies (bins (M/R) r and M/2R + (M/R) r) pair with THEMSELVES: register NB r with NB (R-r), NB r + 1 with
// NB (R-1-r) + 1; bin 0 keeps (A, B) - both real - as the two parts of its one register, bin M/2 (wc = -1: Zy = 4 g2 Z) keeps 4 Z.
template <class PL>
struct UpolsTwiddle {  // wc of the pairs: (cos, sin)(pi t / M) in two registers, (cos, sin)(pi r / R) and (pi (r + 1/2) / R) as literals
    static constexpr int R = PL::RL;
    float c0, s0;
    __device__ __forceinline__ explicit UpolsTwiddle(int tid) { sincospif(static_cast<float>(tid) / static_cast<float>(PL::M), &s0, &c0); }
    __device__ __forceinline__ void regular(int r, float& wr, float& wi) const {  // k = tid + (M/R) r
        constexpr HalfTurn<R> turn = half_turn<R>();
        const float cr = turn.c[r], sr = turn.s[r];
        wr = -fmaf(c0, sr, s0 * cr);  // -sin, -cos of pi k / M
        wi = -fmaf(c0, cr, -s0 * sr);
    }
    static __device__ __forceinline__ void self_even(int r, float& wr, float& wi) {  // thread 0, k = (M/R) r
        constexpr HalfTurn<2 * R> turn = half_turn<2 * R>();
        wr = -turn.s[2 * r];
        wi = -turn.c[2 * r];
    }
    static __device__ __forceinline__ void self_odd(int r, float& wr, float& wi) {  // thread 0, k = M/2R + (M/R) r
        constexpr HalfTurn<2 * R> turn = half_turn<2 * R>();
        wr = -turn.s[2 * r + 1];
        wi = -turn.c[2 * r + 1];
    }
};

__device__ __forceinline__ void upols_split_pair(float& zar, float& zai, float& zbr, float& zbi, float wr, float wi) {
    const float ur = zar + zbr, ui = zai - zbi, dr = zar - zbr, di = zai + zbi;
    const float tr = fmaf(wr, dr, -wi * di), ti = fmaf(wr, di, wi * dr);
    zar = ur + tr;
    zai = ui + ti;
    zbr = ur - tr;
    zbi = ui - ti;
}
__device__ __forceinline__ void upols_merge_pair(float& par, float& pai, float& qbr, float& qbi, float wr, float wi) {
    const float er = par + qbr, ei = pai + qbi, fr = par - qbr, fi = pai - qbi;
    const float orr = fmaf(wr, fr, wi * fi), oi = fmaf(wr, fi, -wi * fr);  // conj(wc) F
    par = er + orr;
    pai = ei + oi;
    qbr = er - orr;
    qbi = oi - ei;
}

// Thread 0's own pairing must not be the other side of an if / else over the 2 P registers: the register allocator keeps what a divergent
// `else` reads alive across the `then` (the compiler's view of a wave is not per lane), which doubles the registers' live ranges (+64: the
// launches spilled, and the 8192-point multiply launch needed 148 registers).  Instead thread 0 parks its registers in `park` (2 P floats of
// LDS behind the exchange buffer), every lane runs the regular pairing - thread 0 on values it will not use - and thread 0 then pairs its
// parked values its own way into the same registers.
// (Two sequential ifs - `tid != 0`, then `tid == 0`, each rewriting in place, what fftconv_core.inc's spectrum_stage does - do not help here: 12 / 32 B of
// scratch again on the 512-thread plan.  Thread 0's second pass must not READ registers.)
template <class PL, class F0, class F1>
__device__ __forceinline__ void upols_over_pairs(float (&xr)[PL::P], float (&xi)[PL::P], int tid, float2* park, F0 pair_fn, F1 bin0_fn) {
    constexpr int R = PL::RL, NB = PL::NBL;
    if (tid == 0) {
#pragma unroll
        for (int m = 0; m < PL::P; ++m) park[m] = make_float2(xr[m], xi[m]);
    }
    const UpolsTwiddle<PL> tw(tid);
    float wr, wi;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        tw.regular(r, wr, wi);
        pair_fn(xr[NB * r], xi[NB * r], xr[NB * (R - 1 - r) + 1], xi[NB * (R - 1 - r) + 1], wr, wi);
    }
    if (tid == 0) {
        auto pair_at = [&](int ma, int mb) {
            float2 a = park[ma], b = park[mb];
            pair_fn(a.x, a.y, b.x, b.y, wr, wi);
            xr[ma] = a.x;
            xi[ma] = a.y;
            xr[mb] = b.x;
            xi[mb] = b.y;
        };
        {
            float2 z0 = park[0], zh = park[NB * (R / 2)];
            bin0_fn(z0.x, z0.y, zh.x, zh.y);
            xr[0] = z0.x;
            xi[0] = z0.y;
            xr[NB * (R / 2)] = zh.x;
            xi[NB * (R / 2)] = zh.y;
        }
#pragma unroll
        for (int r = 1; r < R / 2; ++r) {
            UpolsTwiddle<PL>::self_even(r, wr, wi);
            pair_at(NB * r, NB * (R - r));
        }
#pragma unroll
        for (int r = 0; r < R / 2; ++r) {
            UpolsTwiddle<PL>::self_odd(r, wr, wi);
            pair_at(NB * r + 1, NB * (R - 1 - r) + 1);
        }
    }
}

template <class PL>
__device__ __forceinline__ void upols_split(float (&xr)[PL::P], float (&xi)[PL::P], int tid, float2* park) {
    upols_over_pairs<PL>(
        xr, xi, tid, park, [](float& ar, float& ai, float& br, float& bi, float wr, float wi) { upols_split_pair(ar, ai, br, bi, wr, wi); },
        [](float& z0r, float& z0i, float& zhr, float& zhi) {  // bins 0 and M/2 of thread 0
            const float a = z0r, b = z0i;
            z0r = 2.f * (a + b);  // A = 2 X[0]
            z0i = 2.f * (a - b);  // B = 2 X[M]
            zhr *= 4.f;
            zhi *= 4.f;
        });
}

// `s0`, `s1`: thread 0's sums for bin 0 (real table entries against the real A and B: the parts multiply one by one, not as complex numbers)
template <class PL>
__device__ __forceinline__ void upols_merge(float (&xr)[PL::P], float (&xi)[PL::P], int tid, float2* park, float s0, float s1) {
    upols_over_pairs<PL>(
        xr, xi, tid, park, [](float& ar, float& ai, float& br, float& bi, float wr, float wi) { upols_merge_pair(ar, ai, br, bi, wr, wi); },
        [s0, s1](float& z0r, float& z0i, float&, float&) {  // Zy[0] = (SP + SQ) + i (SP - SQ); bin M/2 is its sum as it stands
            z0r = s0 + s1;
            z0i = s0 - s1;
        });
}

}  // namespace

// ---- launch 1: window -> forward passes -> delay line ----------------------------------------------------------------
template <class PL, bool S16>
__global__ __launch_bounds__(PL::T, PL::P > 32 ? 2 : (PL::T > 256 ? 4 : ADSP_UPOLS_FWD_WAVES)) void upols_forward_kernel(const UpolsArgs a) {
    constexpr int P = PL::P, T = PL::T;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    real2* lds = reinterpret_cast<real2*>(smem_raw);
    const int tid = static_cast<int>(threadIdx.x);
    int c, blk;
    if (!upols_block<PL>(a, c, blk)) return;

    using U = typename std::conditional<S16, unsigned, float>::type;  // storage unit: float, or a dword of two int16
    constexpr int SPU = S16 ? 2 : 1;
    const int N = a.N;
    const size_t plane = static_cast<size_t>(a.C) * N / SPU;
    const size_t chan_units = static_cast<size_t>(c) * N / SPU;
    const bool odd = tid & 1;
    const int t0 = a.rel_first + blk * PL::M + a.nh * N;  // window start on the biased (>= 0) time axis: history chunk -nh is chunk 0

    real xr[P], xi[P];
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    typename std::conditional<S16, v2u, v4f>::type raw[P / 2];
#pragma unroll
    for (int u = 0; u < P / 2; ++u) {
        // even lane: elements (tid, tid+1) of register 2u; odd lane: elements (tid-1, tid) of register 2u+1 (fftconv_core.inc)
        const int elem = (tid - (odd ? 1 : 0)) + T * (2 * u + (odd ? 1 : 0));
        int q, r;
        locate_chunk(t0 + 2 * elem, N, a.inv_n, q, r);
        q -= a.nh;
        const U* base = static_cast<const U*>(a.zeros);
        size_t off = r / SPU;
        if (q < a.n_steps) {
            if (q < 0) {
                int slot = a.ring_pos + 1 + q;
                slot += (slot < 0) ? a.ring_slots : 0;
                slot = slot < 0 ? 0 : slot;
                base = static_cast<const U*>(a.ring) + static_cast<size_t>(slot) * plane;
            } else {
                base = static_cast<const U*>(a.in) + static_cast<size_t>(q) * plane;
            }
            off += chan_units;
        }
        // every sample of a window is read by TWO workgroups (the blocks before and after share a half each): plain loads, so that
        // the neighbour's request is an L2 hit (ADSP_UPOLS_FWD_NT=1: the non-temporal loads of round 5, profiles/r6_upols_forward_loads.txt)
        if constexpr (S16) raw[u] = *reinterpret_cast<const v2u*>(base + off);
        else if constexpr (ADSP_UPOLS_FWD_NT) raw[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(base + off));
        else raw[u] = *reinterpret_cast<const v4f*>(base + off);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S16) {
#pragma unroll
        for (int u = 0; u < P / 2; ++u) {
            const v2u v = raw[u];
            const unsigned sx = lane_xor1_u(odd ? v.x : v.y);
            unpack_s16(odd ? sx : v.x, xr[2 * u], xi[2 * u]);
            unpack_s16(odd ? v.y : sx, xr[2 * u + 1], xi[2 * u + 1]);
        }
    } else {
#pragma unroll
        for (int u = 0; u < P / 2; u += 2) {  // two register pairs per exchange (fftconv_core.inc: lane_pair_exchange4)
            const float ea[4] = {raw[u].x, raw[u].y, raw[u + 1].x, raw[u + 1].y}, eb[4] = {raw[u].z, raw[u].w, raw[u + 1].z, raw[u + 1].w};
            float an[4], bn[4];
            lane_pair_exchange4<(ADSP_DPP_SELECT & 1) && (P < 64 || ADSP_P64_DPP)>(ea, eb, an, bn, odd);
            xr[2 * u] = an[0];
            xi[2 * u] = an[1];
            xr[2 * u + 1] = bn[0];
            xi[2 * u + 1] = bn[1];
            xr[2 * u + 2] = an[2];
            xi[2 * u + 2] = an[3];
            xr[2 * u + 3] = bn[2];
            xi[2 * u + 3] = bn[3];
        }
    }
    int ja, jb;
    upols_indices<PL>(tid, ja, jb);
    run_passes<PL, false, 0, const real4* __restrict__, (PL::P > 32 ? 1 : -1)>(xr, xi, lds, a.tw, tid, ja, jb);  // (64 points per thread: a laundered lane index per exchange, +10 % here - fftconv_core.inc: lane_mode)

    constexpr int RR = PL::RL, NB = PL::NBL;
    __builtin_amdgcn_sched_barrier(0);
    upols_split<PL>(xr, xi, tid, lds + PL::LDS_ELEMS);
    __builtin_amdgcn_sched_barrier(0);

    int slot = a.slot_first + blk;
    slot -= slot >= a.R ? a.R : 0;
    // the delay line holds a block as the multiply kernel reads it: 16 bytes per lane and load - unit 2h = the registers (NB 2h, NB (2h+1))
    // of the paired butterflies' first sides, unit 2h+1 = their partners (NB (R-1-2h) + 1, NB (R-2-2h) + 1); read back by the next launch (L2)
    float4* z = reinterpret_cast<float4*>(a.zline) + (static_cast<size_t>(c) * a.R + slot) * (static_cast<size_t>(P / 2) * T) + tid;
#pragma unroll
    for (int h = 0; h < RR / 2; ++h) {
        const int a0 = NB * (2 * h), a1 = NB * (2 * h + 1), b0 = NB * (RR - 1 - 2 * h) + 1, b1 = NB * (RR - 2 - 2 * h) + 1;
#if defined(ADSP_TUNING_BUILD) && defined(ADSP_UPOLS_Z_NT_STORE)  // tuning A/B: non-temporal stores of the spectrum (profiles/r6_upols_policies.txt)
        typedef float v4f_z __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(v4f_z{xr[a0], xi[a0], xr[a1], xi[a1]}, reinterpret_cast<v4f_z*>(z + static_cast<size_t>(2 * h) * T));
        __builtin_nontemporal_store(v4f_z{xr[b0], xi[b0], xr[b1], xi[b1]}, reinterpret_cast<v4f_z*>(z + static_cast<size_t>(2 * h + 1) * T));
#else
        z[static_cast<size_t>(2 * h) * T] = make_float4(xr[a0], xi[a0], xr[a1], xi[a1]);
        z[static_cast<size_t>(2 * h + 1) * T] = make_float4(xr[b0], xi[b0], xr[b1], xi[b1]);
#endif
    }
}

// The ring keeps what later calls' windows can still reach of this call's input: a window starts less than two blocks before the first
// sample of the call that completes it, so only the LAST 2B samples of a call's input are ever read again - 16384 of Example4's 88200
// per channel.  One workgroup per channel at the end of the multiply launch's grid (which reads neither the input nor the ring)
// copies them, 16 (int16: 8) bytes per lane; rounds 1 - 5 copied whole chunks with hipMemcpyAsync behind the kernels (9 us of a 65 us
// call at 64 channels, 145 of 955 at 1024).
template <bool S16>
__device__ __forceinline__ void upols_keep_tail(const UpolsArgs& a) {
    using V = typename std::conditional<S16, uint2, uint4>::type;  // four samples
    const int c = static_cast<int>(blockIdx.x) - a.mac_grid;
    if (c >= a.C) return;
    const int N = a.N, first = a.n_steps * N - a.tail;
    const size_t plane = static_cast<size_t>(a.C) * N / 4, chan = static_cast<size_t>(c) * N / 4;
    const V* in = static_cast<const V*>(a.in);
    V* ring = static_cast<V*>(a.ring_w);
    const int nt = static_cast<int>(blockDim.x), t0 = static_cast<int>(threadIdx.x);
    // (four requests in flight per lane: one workgroup moves up to 2 B + B samples of its channel, and at 64 channels a call is as long as its longest workgroup)
    // what the previous call's straddling block computed for this call's first carry_n outputs: effect, sample format, the call's output array
    if (a.carry_n > 0) {
        using U = typename std::conditional<S16, unsigned, float>::type;
        constexpr int SPU = S16 ? 2 : 1;
        const float2* cr = a.carry_r + static_cast<size_t>(c) * (a.carry_stride / 2);
        const size_t uplane = static_cast<size_t>(a.C) * N / SPU, uchan = static_cast<size_t>(c) * N / SPU;
        const int len = static_cast<int>(a.epi_p2), n2 = a.carry_n / 2;
        for (int i0 = t0; i0 < n2; i0 += 4 * nt) {
            float2 vv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) vv[j] = i0 + j * nt < n2 ? cr[i0 + j * nt] : make_float2(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * nt;
                if (i < n2) {
                float2 v = vv[j];
                const int tau = 2 * i;  // output times tau, tau + 1 of this call
                int k, r;
                locate_chunk(tau, N, a.inv_n, k, r);
                if (a.epi_op == ADSP_EFFECT_TREMOLO) {  // (the multiply launch's indices: upols_mac_kernel)
                    int n0, n1;
                    if (a.epi_replay) {
                        const int r1 = r + 1 == N ? 0 : r + 1;
                        n0 = (a.epi_phase + r) % len;
                        n1 = (a.epi_phase + r1) % len;
                    } else {
                        n0 = (a.epi_phase + tau) % len;
                        n1 = n0 + 1 == len ? 0 : n0 + 1;
                    }
                    v.x *= tremolo_gain(n0, a.epi_p0, a.epi_p1);
                    v.y *= tremolo_gain(n1, a.epi_p0, a.epi_p1);
                } else if (a.epi_op != 0) {
                    v.x = epilogue_value(v.x, a.epi_op, a.epi_p0, a.epi_p1, a.epi_p2);
                    v.y = epilogue_value(v.y, a.epi_op, a.epi_p0, a.epi_p1, a.epi_p2);
                }
                U* dst = static_cast<U*>(a.out) + static_cast<size_t>(k) * uplane + uchan + r / SPU;
                if constexpr (S16) *dst = pack_s16(v.x, v.y);
                else *reinterpret_cast<float2*>(dst) = v;
                }
            }
        }
    }
    // the input's tail into the ring
    for (int u = t0; u < a.tail / 4; u += nt) {
        int k, r;
        locate_chunk(first + 4 * u, N, a.inv_n, k, r);
        int slot = a.ring_pos + 1 + (k - (a.n_steps - a.cnt));
        slot -= slot >= a.ring_slots ? a.ring_slots : 0;
        ring[static_cast<size_t>(slot) * plane + chan + r / 4] = in[static_cast<size_t>(k) * plane + chan + r / 4];
    }
}

// ---- launch 2: sum over partitions of g_p . S_{b-p} -> re-packing -> inverse passes -> kept half ----------------------------------
template <class PL, bool S16>
__global__ __launch_bounds__(PL::T, PL::P > 32 ? 2 : (PL::T > 256 ? 4 : ADSP_UPOLS_MAC_WAVES)) void upols_mac_kernel(const UpolsArgs a) {
    constexpr int P = PL::P, T = PL::T, R = PL::RL, NB = PL::NBL;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    real2* lds = reinterpret_cast<real2*>(smem_raw);
    const int tid = static_cast<int>(threadIdx.x);
    if (static_cast<int>(blockIdx.x) >= a.mac_grid) {  // workgroup-uniform
        upols_keep_tail<S16>(a);
        return;
    }
    int c, blk;
    if (!upols_block<PL>(a, c, blk)) return;

    real ar[P], ai[P];
#pragma unroll
    for (int m = 0; m < P; ++m) ar[m] = ai[m] = 0.f;
    constexpr size_t kSlot = static_cast<size_t>(P) * T;
    const float2* zc = a.zline + static_cast<size_t>(c) * a.R * kSlot;
    auto block_of = [&](int p) {
        int slot = a.slot_first + blk - p;
        slot += slot < 0 ? a.R : 0;
        slot -= slot >= a.R ? a.R : 0;
        return zc + static_cast<size_t>(slot) * kSlot;
    };

    // One stream of stages - a stage is four registers: two float4 of table, two float4 of spectrum, in the delay line's units (2h: registers
    // NB 2h and NB (2h+1), 2h + 1: NB (R-1-2h) + 1 and NB (R-2-2h) + 1) - with kAhead stages requested ahead of the one being multiplied,
    // across partition boundaries.  Every register is one complex multiply-add per partition (upols_split's header); lane 0's register 0
    // - bin 0, whose two parts are the real A and B - is summed a second time part by part (sp0, sp1: every lane does, lane 0 uses it).
    struct Stage {
        float4 t0, t1;  // g of the stage's four registers: H_p[k] / 4M on first sides, conj(H_p[M-k]) / 4M on their partners (upols_upload_tables)
        float4 za, zb;  // the delay line's units 2h, 2h+1
    };
    // (64 points per thread: 128 accumulators leave room for two stages; 32 points in 512 threads - blocks of 16384 - must stay within 128 registers: one)
    constexpr int kStages = R / 2, kAhead = PL::P > 32 ? 2 : (PL::T > 256 ? ADSP_UPOLS_AHEAD_512 : ADSP_UPOLS_AHEAD);
    static_assert(kStages % kAhead == 0, "the stage ring is indexed at compile time");
    Stage st[kAhead];
    // buffer loads of 16 bytes per lane (the vector-memory path takes ~7 ns per wave instruction whatever its width: micro/tcp_rate.hip):
    // one lane offset, everything else is a scalar offset - 32 global addresses per partition would otherwise be kept in registers
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    auto as_f4 = [](v4u u) { return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)); };
    const int lane16 = tid * 16;
    auto request = [&](Stage& s, const float2* z, const float4* tab, int h) {
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(tab), 0, R * T * 16, 0x00020000);
        const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(z), 0, static_cast<int>(kSlot) * 8, 0x00020000);
#if defined(ADSP_TUNING_BUILD) && (ADSP_UPOLS_ABLATE & 1)  // tuning (wrong results): no table traffic - the bound of sharing table fetches between channels
        s.t0 = s.t1 = make_float4(1.f, 0.f, 0.5f, 0.f);
#else
        s.t0 = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rt, lane16, (2 * h) * T * 16, 0));
        s.t1 = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rt, lane16, (2 * h + 1) * T * 16, 0));
#endif
#if defined(ADSP_TUNING_BUILD) && (ADSP_UPOLS_ABLATE & 2)  // ... no spectrum traffic
        s.za = s.zb = make_float4(0.25f, 0.5f, 0.75f, 1.f);
#else
        s.za = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rz, lane16, (2 * h) * T * 16, 0));
        s.zb = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rz, lane16, (2 * h + 1) * T * 16, 0));
#endif
    };
    float sp0 = 0.f, sp1 = 0.f;
    auto cmac = [](float& xr_, float& xi_, float gr, float gi, float zr, float zi) {
        xr_ = fmaf(-gi, zi, fmaf(gr, zr, xr_));
        xi_ = fmaf(gi, zr, fmaf(gr, zi, xi_));
    };
    auto multiply = [&](const Stage& s, int h) {
        if (h == 0) {
            sp0 = fmaf(s.t0.x, s.za.x, sp0);
            sp1 = fmaf(s.t0.y, s.za.y, sp1);
        }
        cmac(ar[NB * (2 * h)], ai[NB * (2 * h)], s.t0.x, s.t0.y, s.za.x, s.za.y);
        cmac(ar[NB * (2 * h + 1)], ai[NB * (2 * h + 1)], s.t0.z, s.t0.w, s.za.z, s.za.w);
        cmac(ar[NB * (R - 1 - 2 * h) + 1], ai[NB * (R - 1 - 2 * h) + 1], s.t1.x, s.t1.y, s.zb.x, s.zb.y);
        cmac(ar[NB * (R - 2 - 2 * h) + 1], ai[NB * (R - 2 - 2 * h) + 1], s.t1.z, s.t1.w, s.zb.z, s.zb.w);
    };
    {
        // Partition order: block b starts with partition b mod P and wraps around, so that the workgroups of a channel's consecutive
        // blocks - which start together and advance in step - want the SAME block Z_{b-p} of the delay line at the same time: one
        // fetch from HBM serves them all out of the L2.  (In the order 0 .. P-1 they read P different blocks per step and met each
        // again a step later, by when the XCD's other workgroups had pushed 4 - 8 MB through its 4 MB L2: the launch fetched
        // every block about twice, profiles/r5_upols_1024ch_counters.txt.)
        int p = (a.p_first + blk) % a.P;  // = (absolute block index) mod P: the order of a block's sum - and so its rounding - does not depend on the call that computes it
        const float2* z = block_of(p);
        const float4* tab = a.pair + static_cast<size_t>(p) * a.pair_stride;
#pragma unroll
        for (int h = 0; h < kAhead; ++h) request(st[h], z, tab, h);
        auto partition = [&](auto more, const float2* zn, const float4* tabn) {  // straight-line code: no branch inside a partition
#pragma unroll
            for (int h = 0; h < kStages; ++h) {
                multiply(st[h % kAhead], h);
                if (h + kAhead < kStages) request(st[h % kAhead], z, tab, h + kAhead);
                else if constexpr (decltype(more)::value) request(st[h % kAhead], zn, tabn, h + kAhead - kStages);
                __builtin_amdgcn_sched_barrier(0);  // keep the requests where they are: kAhead stages of registers, not a partition's
            }
        };
        for (int s = 1; s < a.P; ++s) {
            p = p + 1 == a.P ? 0 : p + 1;
            const float2* zn = block_of(p);
            const float4* tabn = a.pair + static_cast<size_t>(p) * a.pair_stride;
            partition(std::true_type{}, zn, tabn);
            z = zn;
            tab = tabn;
        }
        partition(std::false_type{}, z, tab);
    }

    __builtin_amdgcn_sched_barrier(0);  // (the inverse passes' address arithmetic and twiddle requests stay behind the re-packing: hoisted above it they spill)
    upols_merge<PL>(ar, ai, tid, lds + PL::LDS_ELEMS, sp0, sp1);
    __builtin_amdgcn_sched_barrier(0);
    int ja, jb;
    upols_indices<PL>(tid, ja, jb);
    run_passes<PL, true, 0, const real4* __restrict__, (PL::P > 32 ? 1 : -1)>(ai, ar, lds, a.tw, tid, ja, jb);  // inverse = forward on swapped parts

    if (a.carry_w && blk == a.nblk - 1) {  // workgroup-uniform: the launch's last block reaches beyond the call - those samples are the next call's first outputs
        const int total_c = a.n_steps * a.N;
        const int tau_c = a.rel_first + blk * PL::M - PL::M + 2 * tid;
        float2* cw = a.carry_w + static_cast<size_t>(c) * (PL::M / 2);
#pragma unroll
        for (int m = P / 2; m < P; ++m) {
            const int tau = tau_c + 2 * T * m;
            if (tau >= total_c) cw[(tau - total_c) >> 1] = make_float2(ar[m], ai[m]);
        }
    }

    if (a.epi_op == ADSP_EFFECT_TREMOLO) {
        // the LFO's time base is the stream's own: register m of thread tid holds output times tau, tau + 1 with
        // tau = rel_first + (blk - 1) M + 2 (tid + T m) relative to the call's first sample, whose table index is epi_phase -
        // the same for every channel, as the reference's one-device-per-channel loop has it (EffectTremolo.py:40-46)
        const int len = static_cast<int>(a.epi_p2);
        const float inv_len = 1.f / a.epi_p2, inv_n = a.inv_n;
        const int tau0 = a.rel_first + blk * PL::M - PL::M + 2 * tid;
        if (a.epi_replay) {
            int r0 = tau0 % a.N;  // (negative for samples that are not stored)
            r0 += r0 < 0 ? a.N : 0;
#pragma unroll
            for (int m = P / 2; m < P; ++m) {
                const int r = small_mod(r0 + 2 * T * m, a.N, inv_n);
                const int r1 = r + 1 == a.N ? 0 : r + 1;
                ar[m] *= tremolo_gain(small_mod(a.epi_phase + r, len, inv_len), a.epi_p0, a.epi_p1);
                ai[m] *= tremolo_gain(small_mod(a.epi_phase + r1, len, inv_len), a.epi_p0, a.epi_p1);
            }
        } else {
            int base = (a.epi_phase + tau0) % len;
            base += base < 0 ? len : 0;
#pragma unroll
            for (int m = P / 2; m < P; ++m) {
                const int n = small_mod(base + 2 * T * m, len, inv_len);
                const int n1 = n + 1 == len ? 0 : n + 1;
                ar[m] *= tremolo_gain(n, a.epi_p0, a.epi_p1);
                ai[m] *= tremolo_gain(n1, a.epi_p0, a.epi_p1);
            }
        }
    } else if (a.epi_op != 0) {  // wave-uniform: a stateless effect on the kept half (Saturator, SoftClipper, HardDistortion, Volume, BitCrusher)
#pragma unroll
        for (int m = P / 2; m < P; ++m) {
            ar[m] = epilogue_value(ar[m], a.epi_op, a.epi_p0, a.epi_p1, a.epi_p2);
            ai[m] = epilogue_value(ai[m], a.epi_op, a.epi_p0, a.epi_p1, a.epi_p2);
        }
    }

    // kept: circular indices [B, 2B) = elements >= M/2 = registers m >= P/2; register m holds output time o + 2 (tid + T m) - B
    using U = typename std::conditional<S16, unsigned, float>::type;
    constexpr int SPU = S16 ? 2 : 1;
    const int N = a.N;
    const size_t plane = static_cast<size_t>(a.C) * N / SPU;
    const size_t chan_units = static_cast<size_t>(c) * N / SPU;
    const bool odd = tid & 1;
    const int total = a.n_steps * N;
    const int o = a.rel_first + blk * PL::M - PL::M;  // output time of circular index 0 of this block
    float ea[4] = {0.f, 0.f, 0.f, 0.f}, eb[4] = {0.f, 0.f, 0.f, 0.f};  // (float32: the lane pairs' 16 bytes of two register pairs, lane_pair_exchange4)
    static_assert((P / 4) % 2 == 0, "register pairs are exchanged two at a time");
#pragma unroll
    for (int u = P / 4; u < P / 2; ++u) {
        const int elem = (tid - (odd ? 1 : 0)) + T * (2 * u + (odd ? 1 : 0));
        const int tau = o + 2 * elem;
        unsigned w0 = 0, w1 = 0, swx = 0;
        if constexpr (S16) {
            w0 = pack_s16(ar[2 * u], ai[2 * u]);
            w1 = pack_s16(ar[2 * u + 1], ai[2 * u + 1]);
            swx = lane_xor1_u(odd ? w0 : w1);
        } else if ((u & 1) == 0) {
            const float xa[4] = {ar[2 * u], ai[2 * u], ar[2 * u + 2], ai[2 * u + 2]}, xb[4] = {ar[2 * u + 1], ai[2 * u + 1], ar[2 * u + 3], ai[2 * u + 3]};
            lane_pair_exchange4<(ADSP_DPP_SELECT & 2) && (P < 64 || ADSP_P64_DPP)>(xa, xb, ea, eb, odd);
        }
        if (tau >= 0 && tau < total) {  // (the ends of a call cut a block: multiples of 4 samples on both sides)
            int k, r;
            locate_chunk(tau, N, a.inv_n, k, r);
            U* dst = static_cast<U*>(a.out) + static_cast<size_t>(k) * plane + chan_units + r / SPU;
            if constexpr (S16) {
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
                const v2u v = odd ? v2u{swx, w1} : v2u{w0, swx};
                __builtin_nontemporal_store(v, reinterpret_cast<v2u*>(dst));
            } else {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f v = v4f{ea[2 * (u & 1)], ea[2 * (u & 1) + 1], eb[2 * (u & 1)], eb[2 * (u & 1) + 1]};
                __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(dst));
            }
        }
    }
}

}  // namespace adsp

// ------------------------------------------------------------------------------------------------------------------------------
namespace {
using namespace adsp;
using namespace adsp::tables;
// The block sizes of this build: B = 8192 on the 32-points-per-thread plan (32 KiB of LDS, three / two workgroups per CU) and
// B = 16384 on 32 points per thread in 512 threads (64 KiB, two workgroups of eight waves per CU).  Per output sample the multiply launch reads
// n_partitions x 16 bytes (8 of table, 8 of spectrum), and n_partitions = ceil(taps / B): the larger block halves what bounds the engine, for ~8 % more transform work.
struct UpolsPlan {
    int block, threads, lds_bytes;
    PlanInfo shape;
    const void* fwd[2];  // [sample format: f32, s16]
    const void* mac[2];
};

template <class PL>
UpolsPlan upols_plan() {
    UpolsPlan p;
    p.block = PL::M;
    p.threads = PL::T;
    p.lds_bytes = (PL::LDS_ELEMS + PL::P) * (int)sizeof(float2);  // the exchange buffer + thread 0's parked registers (upols_over_pairs)
    p.shape = PlanInfo{PL::M, 8, PL::P, PL::T, 1, PL::NP, PL::XL ? 1 : 0, {PL::fwd(0), PL::fwd(1), PL::fwd(2), PL::fwd(3)},
                       PL::tw_total, PL::LDS_ELEMS * (int)sizeof(float2), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    p.fwd[0] = reinterpret_cast<const void*>(&upols_forward_kernel<PL, false>);
    p.fwd[1] = reinterpret_cast<const void*>(&upols_forward_kernel<PL, true>);
    p.mac[0] = reinterpret_cast<const void*>(&upols_mac_kernel<PL, false>);
    p.mac[1] = reinterpret_cast<const void*>(&upols_mac_kernel<PL, true>);
    return p;
}

// Blocks of 16384 run as 32 points per thread in 512 threads (what the single-transform engines run since round 6, plan_table.hpp), one stage of
// requests ahead: 125 / 122 registers without scratch, four waves per SIMD.  Against 64 points per thread in 256 threads (two waves per SIMD, two
// stages ahead; rounds 5 - 6a) a call takes -15 % / -9.5 % at 256 channels and -5 ... -7 % / -4.3 % at 1024 (profiles/r6f_upols_split_spectra_ab.txt,
// session 35).  It had LOST 9 - 11 % there while the multiply launch still multiplied unsplit spectra (profiles/r6_upols_block_16384_512_threads.txt)
// and it spilled until thread 0's pairing stopped being the other side of an if / else (upols_over_pairs).  -DADSP_UPOLS_PLAN_16384=ADSP_PLAN_16384_64PT
// builds the earlier form.
#ifndef ADSP_UPOLS_PLAN_16384
#define ADSP_UPOLS_PLAN_16384 ADSP_PLAN_16384
#endif
const UpolsPlan* upols_plans(int* count) {
    static const UpolsPlan plans[] = {upols_plan<ADSP_PLAN_8192>(), upols_plan<ADSP_UPOLS_PLAN_16384>()};
    if (count) *count = (int)(sizeof plans / sizeof plans[0]);
    return plans;
}
}  // namespace

struct adsp_upols {
    adsp_upols_config cfg;
    const UpolsPlan* plan;
    int nh, ring_slots, ring_pos, R;
    long long steps_done;  // chunks consumed so far: the call's first new sample has absolute index steps_done * N
    long long fwd_done;    // every block <= this one has been transformed (-1 at stream start: blocks before it are all zeros)
    char* ring;
    char* zeros;
    float4* tw;
    float4* pair;
    float2* zline;
    float2* carry[2];  // [C][B/2] each: what the last block of a call computed beyond the call's end (upols_launch_pair), written alternately
    int carry_w;       // the buffer the next launch writes
    int carry_len;     // samples the previous call left for the head of the next one (multiple of 4, < B)
    bool carry_valid;  // false: nothing usable (first call, reset, restored state, new filter) - the launch starts with the straddling block
    int carry_mode;    // adsp_upols_set_carry: -1 the library decides per call, 0 never, 1 always
    int cus;           // compute units of the device
    int pair_stride;
    int epi_op;
    float epi_p[3];
    int lfo_len;             // fused tremolo: LFO table length and the length of the reference's LFO buffer (EffectTremolo.py:40-45): the
    long long lfo_copy_len;  // effect's whole state
    int epi_phase, epi_replay;
    bool prepared;
    char *stage_in, *stage_out;
    size_t stage_bytes;
    std::vector<float>* spectra;  // host copy of the [P][B + 1] interleaved partition spectra the tables were built from (get / broadcast)
    float* d_spectra;          // the same on the device, behind a 16-float header: the buffer of the RCCL broadcasts (allocated on first use)
    hipEvent_t ev_done;        // recorded behind every launch pair: a call on ANOTHER stream waits for it (the delay line and the ring are shared state)
    hipStream_t last_stream;
    bool launched;
    size_t ssize() const { return cfg.sample_format == ADSP_FORMAT_F32 ? sizeof(float) : sizeof(short); }
    size_t plane_bytes() const { return (size_t)cfg.n_channels * cfg.chunk_size * ssize(); }
    size_t zline_bytes() const { return (size_t)cfg.n_channels * R * plan->block * sizeof(float2); }
};

namespace {
long long floor_div(long long a, long long b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// adsp_capi.hip's tremolo_run for this engine: the reference's tremolo keeps a buffer of LFO tables and cuts each chunk off its front
// (EffectTremolo.py:40-45); its length is the whole state, the next chunk starts at table index (-length) mod table; when the buffer
// holds EXACTLY one chunk, `copy[-0:]` keeps all of it and every later chunk replays that segment.  Sets the phase / replay flag of
// the next launch pair and returns how many of the next max_steps chunks run on contiguously.
int upols_tremolo_run(adsp_upols* u, int max_steps) {
    const long long N = u->cfg.chunk_size, L = u->lfo_len;
    long long len = u->lfo_copy_len;
    u->epi_replay = 0;
    if (len == N) {
        u->epi_replay = 1;
        u->epi_phase = (int)((L - N % L) % L);
        return max_steps;
    }
    while (len < N) len += L;
    u->epi_phase = (int)((L - len % L) % L);
    int run = 0;
    while (run < max_steps) {
        while (len < N) len += L;
        ++run;
        if (len == N) break;
        len -= N;
    }
    u->lfo_copy_len = len;
    return run;
}

int upols_launch_pair(adsp_upols* u, const void* d_in, void* d_out, int n, hipStream_t stream) {
    const adsp_upols_config& c = u->cfg;
    const long long N = c.chunk_size, t_call = u->steps_done * N, t_end = t_call + (long long)n * N;
    const UpolsPlan& pl = *u->plan;
    const long long kB = pl.block;
    void* kargs[] = {nullptr};
    UpolsArgs a;
    kargs[0] = &a;
    memset(&a, 0, sizeof a);
    a.ring = u->ring;
    a.in = d_in;
    a.out = d_out;
    a.zeros = u->zeros;
    a.tw = u->tw;
    a.pair = u->pair;
    a.zline = u->zline;
    a.ring_pos = u->ring_pos;
    a.ring_slots = u->ring_slots;
    a.C = c.n_channels;
    a.N = c.chunk_size;
    a.nh = u->nh;
    a.n_steps = n;
    a.inv_n = 1.0f / (float)c.chunk_size;
    a.P = c.n_partitions;
    a.R = u->R;
    a.ncg = c.n_channels;
    a.pair_stride = u->pair_stride;
    a.epi_op = u->epi_op;
    a.epi_p0 = u->epi_p[0];
    a.epi_p1 = u->epi_p[1];
    a.epi_p2 = u->epi_p[2];
    a.epi_phase = u->epi_phase;
    a.epi_replay = u->epi_replay;
    const long long groups = ((long long)c.n_channels + 7) / 8 * 8;
    const bool s16 = c.sample_format != ADSP_FORMAT_F32;
    // 1. forward: the blocks whose 2B-sample windows [(b-1)B, (b+1)B) this call completes
    const long long b_fwd_hi = floor_div(t_end, kB) - 1;
    if (b_fwd_hi > u->fwd_done) {
        const long long b0 = u->fwd_done + 1;
        a.nblk = (int)(b_fwd_hi - b0 + 1);
        a.slot_first = (int)(((b0 % u->R) + u->R) % u->R);
        a.rel_first = (int)((b0 - 1) * kB - t_call);
        if ((long long)a.rel_first + (long long)u->nh * N < 0) return fail(ADSP_ERR_STATE, "internal: block %lld starts before the input history", b0);
        const long long grid = groups * a.nblk;
        if (grid > 0x7fffffffLL) return fail(ADSP_ERR_ARG, "launch too large (%lld workgroups)", grid);
        HIP_TRY(hipLaunchKernel(pl.fwd[s16 ? 1 : 0], dim3((unsigned)grid), dim3(pl.threads), kargs, pl.lds_bytes, stream));
        u->fwd_done = b_fwd_hi;
    }
    // 2. multiply-accumulate + inverse: the blocks of y that meet this call's outputs, y index tau - delay for tau in [t_call, t_end)
    const long long b_lo = floor_div(t_call - c.delay, kB), b_hi = floor_div(t_end - c.delay - 1, kB);
    if (b_hi > u->fwd_done) return fail(ADSP_ERR_STATE, "internal: output block %lld needs input that has not arrived", b_hi);
    if (u->fwd_done - (b_lo - c.n_partitions + 1) >= u->R) return fail(ADSP_ERR_STATE, "internal: the delay line is too short for this call");
    // Block b_lo straddles the start of the call whenever t_call - delay is not a multiple of B: the PREVIOUS call computed it whole (its inputs had
    // arrived: delay >= B), stored its own part and left the rest - this call's first carry_len outputs - in a carry buffer, which this launch's
    // per-channel workgroups copy out (rounds 5 - 6b multiplied and inverse-transformed such a block in both calls: one block in 6.4 at B = 16384
    // and Example4's chunk).  Not when nothing usable was left, or when the call ends inside that very block.
    // ... and only where it pays: with fewer than two workgroups per CU a call is as long as its longest workgroup, and the per-channel workgroup's
    // extra copy makes it 1 - 2 % longer (32 / 64 channels of Example4's chunk); from 128 channels on a call takes -2 ... -7 %
    // (profiles/r6f_upols_carry_ab.txt).  A call that does not carry leaves nothing for the next one either.
    const bool carry_on = u->carry_mode > 0 || (u->carry_mode < 0 && groups * (b_hi - b_lo + 1) >= 2LL * u->cus);
    const bool use_carry = carry_on && u->carry_valid && u->carry_len > 0 && b_hi >= b_lo + 1;
    const long long b_first = use_carry ? b_lo + 1 : b_lo;
    a.nblk = (int)(b_hi - b_first + 1);
    a.slot_first = (int)(((b_first % u->R) + u->R) % u->R);
    a.rel_first = (int)(b_first * kB + c.delay - t_call);  // output time of the block's first kept sample (circular index B)
    a.p_first = (int)(((b_first % c.n_partitions) + c.n_partitions) % c.n_partitions);
    a.carry_w = carry_on ? u->carry[u->carry_w] : nullptr;
    a.carry_r = u->carry[u->carry_w ^ 1];
    a.carry_stride = (int)kB;
    a.carry_n = use_carry ? u->carry_len : 0;
    const long long grid = groups * a.nblk;
    // ... plus one workgroup per channel that keeps the input's tail (the last min(2B, n N) samples) in the ring for the next call's windows
    const int cnt = n < u->nh ? n : u->nh;
    a.ring_w = u->ring;
    a.mac_grid = (int)grid;
    a.tail = (int)((long long)n * N < 2 * kB ? (long long)n * N : 2 * kB);
    a.cnt = cnt;
    if (grid + c.n_channels > 0x7fffffffLL) return fail(ADSP_ERR_ARG, "launch too large (%lld workgroups)", grid + c.n_channels);
    HIP_TRY(hipLaunchKernel(pl.mac[s16 ? 1 : 0], dim3((unsigned)(grid + c.n_channels)), dim3(pl.threads), kargs, pl.lds_bytes, stream));
    u->ring_pos = (u->ring_pos + cnt) % u->ring_slots;
    u->steps_done += n;
    u->carry_len = (int)((b_hi + 1) * kB + c.delay - t_end);  // what block b_hi holds beyond this call: 0 .. B-4
    u->carry_valid = carry_on;
    u->carry_w ^= 1;
    return ADSP_OK;
}
}  // namespace


namespace {
size_t upols_spectra_floats(const adsp_upols* u) { return (size_t)u->cfg.n_partitions * 2 * ((size_t)u->plan->block + 1); }

// Host spectra -> the multiply launch's tables (allocated on first use; later calls - adsp_upols_set_spectra, a broadcast - overwrite
// them: the caller has drained the device).  Per partition one table in the delay line's layout, [R][T] float4: unit 2h = the registers
// (NB 2h, NB (2h+1)) of thread t, unit 2h + 1 = (NB (R-1-2h) + 1, NB (R-2-2h) + 1).  Register NB r + i holds bin j_i + (M/R) r of the
// butterfly j_0 = t, j_1 = M/R - t (thread 0: 0 and M/2R); a register that is the FIRST side of its pair carries g1 = H[bin] / 4M, a
// partner g2 = conj(H[bin]) / 4M (upols_split's header).  Thread 0's butterflies pair with themselves: their upper halves are the
// partners; bin M/2 pairs with itself and carries g2; bin 0's register holds the real (A, B) and carries (H[0], H[M]) / 4M as its two
// parts - the imaginary parts of the kernel's bins 0 and M are zero (a real kernel) and are not looked at.
int upols_upload_tables(adsp_upols* u, const float* spectra) {
    const adsp_upols_config* cfg = &u->cfg;
    const PlanInfo pl = u->plan->shape;
    const int kB = u->plan->block;
    std::vector<float4> all;
    const int RR = pl.rad[pl.NP - 1], D = kB / RR, T = pl.T, NB = pl.P / RR;
    if (pl.XL || NB != 2) return fail(ADSP_ERR_STATE, "internal: the partitioned engines run in-register pairing plans with one pair of butterflies per thread");
    u->pair_stride = RR * T;
    all.resize((size_t)cfg->n_partitions * u->pair_stride);
    const double sc = 1.0 / (4.0 * (double)kB);
    for (int p = 0; p < cfg->n_partitions; ++p) {
        const float* H = spectra + (size_t)p * 2 * (kB + 1);
        auto entry = [&](int t, int m, float& gr, float& gi) {  // register m of thread t
            const int i = m % NB, r = m / NB;
            const int bin = (i == 0 ? t : (t == 0 ? D / 2 : D - t)) + D * r;
            const bool partner = t == 0 ? r >= RR / 2 : i == 1;
            if (t == 0 && m == 0) {
                gr = (float)((double)H[0] * sc);
                gi = (float)((double)H[2 * kB] * sc);
                return;
            }
            gr = (float)((double)H[2 * bin] * sc);
            gi = (float)((partner ? -1.0 : 1.0) * (double)H[2 * bin + 1] * sc);
        };
        for (int h = 0; h < RR / 2; ++h)
            for (int t = 0; t < T; ++t) {
                float4 a, b;
                entry(t, NB * (2 * h), a.x, a.y);
                entry(t, NB * (2 * h + 1), a.z, a.w);
                entry(t, NB * (RR - 1 - 2 * h) + 1, b.x, b.y);
                entry(t, NB * (RR - 2 - 2 * h) + 1, b.z, b.w);
                all[(size_t)p * u->pair_stride + (size_t)(2 * h) * T + t] = a;
                all[(size_t)p * u->pair_stride + (size_t)(2 * h + 1) * T + t] = b;
            }
    }
    if (!u->pair) HIP_TRY(hipMalloc(&u->pair, all.size() * sizeof(float4)));
    HIP_TRY(hipMemcpy(u->pair, all.data(), all.size() * sizeof(float4), hipMemcpyHostToDevice));
    u->carry_valid = false;  // (a carried block was multiplied by the filter before)
    if (!u->spectra) u->spectra = new std::vector<float>();
    u->spectra->assign(spectra, spectra + upols_spectra_floats(u));
    return ADSP_OK;
}
}  // namespace

extern "C" {

int adsp_upols_block_size(void) { return upols_plans(nullptr)[0].block; }

int adsp_upols_block_sizes(int* sizes, int capacity) {
    int n = 0;
    const UpolsPlan* plans = upols_plans(&n);
    for (int i = 0; i < n && sizes && i < capacity; ++i) sizes[i] = plans[i].block;
    return n;
}

int adsp_upols_create(const adsp_upols_config* cfg, const float* spectra, adsp_upols** out) {
    if (!cfg || !spectra || !out) return fail(ADSP_ERR_ARG, "NULL argument");
    *out = nullptr;
    int n_plans = 0;
    const UpolsPlan* plans = upols_plans(&n_plans);
    const UpolsPlan* plan = nullptr;
    for (int i = 0; i < n_plans; ++i)
        if (plans[i].block == cfg->block_size) plan = &plans[i];
    if (!plan) return fail(ADSP_ERR_ARG, "block_size %d: this build partitions into blocks of %d or %d samples (adsp_upols_block_sizes)", cfg->block_size, plans[0].block, plans[n_plans - 1].block);
    const int kB = plan->block;
    if (cfg->chunk_size < 16 || cfg->chunk_size % 4) return fail(ADSP_ERR_ARG, "chunk_size %d: partitioned engines need a multiple of 4, >= 16", cfg->chunk_size);
    if (cfg->n_channels <= 0) return fail(ADSP_ERR_ARG, "n_channels must be positive");
    if (cfg->n_partitions < 1 || cfg->n_partitions > 4096) return fail(ADSP_ERR_ARG, "n_partitions %d out of range 1..4096", cfg->n_partitions);
    if (cfg->delay < kB || cfg->delay % 4)
        return fail(ADSP_ERR_ARG, "delay %d: must be a multiple of 4 and >= the block size %d (an output block may only need input that has arrived)", cfg->delay, kB);
    if (cfg->sample_format != ADSP_FORMAT_F32 && cfg->sample_format != ADSP_FORMAT_S16)
        return fail(ADSP_ERR_ARG, "sample_format %d: ADSP_FORMAT_F32 or ADSP_FORMAT_S16", cfg->sample_format);
    if (cfg->max_steps < 1) return fail(ADSP_ERR_ARG, "max_steps must be positive");
    const long long span = (long long)cfg->max_steps * cfg->chunk_size + cfg->delay + 4LL * kB;
    if (span >= 0x7fffffffLL) return fail(ADSP_ERR_ARG, "max_steps %d x chunk %d + delay is too long for one launch", cfg->max_steps, cfg->chunk_size);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(ADSP_ERR_NO_DEVICE, "no HIP device");
    }
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(ADSP_ERR_ARG, "device_id %d out of range (%d devices)", cfg->device_id, ndev);
    adsp_upols* u = new adsp_upols();
    memset(static_cast<void*>(u), 0, sizeof *u);
    u->cfg = *cfg;
    u->plan = plan;
    const int N = cfg->chunk_size;
    u->nh = (2 * kB + N - 1) / N;  // the oldest window a call opens starts less than two blocks before the call's first sample
    u->ring_slots = u->nh + 1;
    u->ring_pos = u->ring_slots - 1;
    // blocks b_lo - P + 1 .. fwd_done of a call must be distinct slots: see upols_launch_pair
    u->R = (int)(((long long)cfg->max_steps * N + cfg->delay + kB - 1) / kB) + cfg->n_partitions + 3;
    u->steps_done = 0;
    u->fwd_done = -1;
    u->carry_mode = -1;
    u->cus = 256;
    auto bail = [&](int code) {
        adsp_upols_destroy(u);
        return code;
    };
    hipError_t err;
    if ((err = hipSetDevice(cfg->device_id)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(err)));
    if ((err = hipDeviceGetAttribute(&u->cus, hipDeviceAttributeMultiprocessorCount, cfg->device_id)) != hipSuccess || u->cus <= 0) u->cus = 256;
    if ((err = hipEventCreateWithFlags(&u->ev_done, hipEventDisableTiming)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(err)));
    for (const void* fn : {plan->fwd[0], plan->fwd[1], plan->mac[0], plan->mac[1]})
        if ((err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, plan->lds_bytes)) != hipSuccess)
            return bail(fail(ADSP_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(err)));
    const size_t ring_bytes = (size_t)u->ring_slots * u->plane_bytes();
    if ((err = hipMalloc(&u->ring, ring_bytes)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMalloc ring (%zu bytes): %s", ring_bytes, hipGetErrorString(err)));
    if ((err = hipMemset(u->ring, 0, ring_bytes)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMemset: %s", hipGetErrorString(err)));
    if ((err = hipMalloc(&u->zeros, (size_t)N * sizeof(float))) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMalloc: %s", hipGetErrorString(err)));
    if ((err = hipMemset(u->zeros, 0, (size_t)N * sizeof(float))) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMemset: %s", hipGetErrorString(err)));
    if ((err = hipMalloc(&u->zline, u->zline_bytes())) != hipSuccess)
        return bail(fail(ADSP_ERR_HIP, "hipMalloc delay line (%zu bytes = %d channels x %d blocks x %d bytes): %s", u->zline_bytes(), cfg->n_channels, u->R,
                         (int)(kB * sizeof(float2)), hipGetErrorString(err)));
    if ((err = hipMemset(u->zline, 0, u->zline_bytes())) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMemset: %s", hipGetErrorString(err)));
    for (int i = 0; i < 2; ++i)
        if ((err = hipMalloc(&u->carry[i], (size_t)cfg->n_channels * kB * sizeof(float))) != hipSuccess)
            return bail(fail(ADSP_ERR_HIP, "hipMalloc carry buffer (%zu bytes): %s", (size_t)cfg->n_channels * kB * sizeof(float), hipGetErrorString(err)));
    // tables: the plan's twiddles, and per partition the pair tables of its spectrum - built exactly like an engine's
    const PlanInfo pl = plan->shape;
    std::vector<float4> tw;
    build_twiddles<float>(pl, tw);
    if ((int)tw.size() != pl.tw_total) return bail(fail(ADSP_ERR_STATE, "internal: twiddle count %zu != %d", tw.size(), pl.tw_total));
    tw.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
    if ((err = hipMalloc(&u->tw, tw.size() * sizeof(float4))) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMalloc: %s", hipGetErrorString(err)));
    if ((err = hipMemcpy(u->tw, tw.data(), tw.size() * sizeof(float4), hipMemcpyHostToDevice)) != hipSuccess)
        return bail(fail(ADSP_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(err)));
    const int rc_tab = upols_upload_tables(u, spectra);
    if (rc_tab) return bail(rc_tab);
    *out = u;
    return ADSP_OK;
}

void adsp_upols_destroy(adsp_upols* u) {
    if (!u) return;
    (void)hipSetDevice(u->cfg.device_id);
    (void)hipDeviceSynchronize();
    for (void* p : {(void*)u->ring, (void*)u->zeros, (void*)u->tw, (void*)u->pair, (void*)u->zline, (void*)u->carry[0], (void*)u->carry[1], (void*)u->stage_in, (void*)u->stage_out})
        if (p) (void)hipFree(p);
    if (u->ev_done) (void)hipEventDestroy(u->ev_done);
    if (u->d_spectra) (void)hipFree(u->d_spectra);
    delete u->spectra;
    delete u;
}

int adsp_upols_reset(adsp_upols* u) {
    if (!u) return fail(ADSP_ERR_ARG, "NULL engine");
    HIP_TRY(hipSetDevice(u->cfg.device_id));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(u->ring, 0, (size_t)u->ring_slots * u->plane_bytes()));
    HIP_TRY(hipMemset(u->zline, 0, u->zline_bytes()));
    u->ring_pos = u->ring_slots - 1;
    u->steps_done = 0;
    u->fwd_done = -1;
    u->carry_valid = false;
    u->lfo_copy_len = u->lfo_len;  // ... and the fused tremolo's LFO restarts (EffectTremolo.py:49-57)
    u->epi_phase = u->epi_replay = 0;
    return ADSP_OK;
}

int adsp_upols_set_epilogue(adsp_upols* u, int effect, float p0, float p1, float p2) {
    if (!u) return fail(ADSP_ERR_ARG, "NULL engine");
    if (effect < ADSP_EFFECT_NONE || effect > ADSP_EFFECT_BIT_CRUSHER) return fail(ADSP_ERR_ARG, "unknown effect %d", effect);
    if (effect != ADSP_EFFECT_NONE && u->cfg.sample_format != ADSP_FORMAT_F32) return fail(ADSP_ERR_ARG, "fused effects need a float32 engine");
    if (effect == ADSP_EFFECT_TREMOLO && !(p2 >= 1.f && p2 <= 8388608.f)) return fail(ADSP_ERR_ARG, "tremolo: p2 = LFO table length in samples (1..2^23)");
    u->epi_op = effect;
    u->epi_p[0] = p0;
    u->epi_p[1] = p1;
    u->epi_p[2] = p2;
    u->lfo_len = effect == ADSP_EFFECT_TREMOLO ? (int)p2 : 0;
    u->lfo_copy_len = u->lfo_len;  // a fresh LFO: one table in the buffer (EffectTremolo.py:24)
    u->epi_phase = u->epi_replay = 0;
    return ADSP_OK;
}

int adsp_upols_info(const adsp_upols* u, int* history_chunks, int* delay_line_blocks, size_t* delay_line_bytes) {
    if (!u) return fail(ADSP_ERR_ARG, "NULL engine");
    if (history_chunks) *history_chunks = u->nh;
    if (delay_line_blocks) *delay_line_blocks = u->R;
    if (delay_line_bytes) *delay_line_bytes = u->zline_bytes();
    return ADSP_OK;
}

int adsp_upols_apply_device(adsp_upols* u, const void* d_in, void* d_out, int n_steps, void* stream) {
    if (!u || !d_in || !d_out) return fail(ADSP_ERR_ARG, "NULL argument");
    if (n_steps <= 0) return fail(ADSP_ERR_ARG, "n_steps must be positive");
    HIP_TRY(hipSetDevice(u->cfg.device_id));
    const size_t plane = u->plane_bytes();
    {   // the second launch of a pair writes outputs while later blocks of the call still read their input windows
        const char *i0 = static_cast<const char*>(d_in), *o0 = static_cast<const char*>(d_out);
        const size_t span = (size_t)n_steps * plane;
        if (i0 < o0 + span && o0 < i0 + span) return fail(ADSP_ERR_ARG, "d_in and d_out overlap: the partitioned engines do not run in place");
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (u->launched && st != u->last_stream) HIP_TRY(hipStreamWaitEvent(st, u->ev_done, 0));  // the previous call ran on another stream
    for (int done = 0; done < n_steps;) {  // at most max_steps chunks per launch pair (the delay line is sized for that)
        int n = n_steps - done < u->cfg.max_steps ? n_steps - done : u->cfg.max_steps;
        if (u->epi_op == ADSP_EFFECT_TREMOLO) n = upols_tremolo_run(u, n);  // the LFO runs on contiguously except where the reference's buffer quirk restarts it
        const int rc = upols_launch_pair(u, static_cast<const char*>(d_in) + (size_t)done * plane, static_cast<char*>(d_out) + (size_t)done * plane, n, st);
        if (rc) return rc;
        done += n;
    }
    HIP_TRY(hipEventRecord(u->ev_done, st));
    u->last_stream = st;
    u->launched = true;
    return ADSP_OK;
}

int adsp_upols_set_carry(adsp_upols* u, int mode) {
    if (!u) return fail(ADSP_ERR_ARG, "NULL engine");
    if (mode < -1 || mode > 1) return fail(ADSP_ERR_ARG, "mode %d: -1 (automatic), 0 (never) or 1 (always)", mode);
    u->carry_mode = mode;  // (what a previous call carried stays usable: the same samples either way)
    return ADSP_OK;
}

int adsp_upols_synchronize(adsp_upols* u, void* stream) {
    if (!u) return fail(ADSP_ERR_ARG, "NULL engine");
    HIP_TRY(hipSetDevice(u->cfg.device_id));
    if (u->launched && static_cast<hipStream_t>(stream) != u->last_stream) HIP_TRY(hipEventSynchronize(u->ev_done));
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return ADSP_OK;
}

// A checkpoint is the engine's whole device state: counters, the input ring (tails) and the frequency-domain delay line.
namespace {
struct UpolsStateHeader {
    unsigned magic, version;
    adsp_upols_config cfg;
    int ring_slots, R, ring_pos, reserved;
    long long steps_done, fwd_done, lfo_copy_len;
    unsigned long long ring_bytes, zline_bytes;
};
constexpr unsigned kStateMagic = 0x55504f32u;  // "UPO2": the delay line holds split half spectra ("UPOL" = 0x55504f4c: the unsplit Z of the builds before)
}  // namespace

int adsp_upols_state_bytes(const adsp_upols* u, size_t* bytes) {
    if (!u || !bytes) return fail(ADSP_ERR_ARG, "NULL argument");
    *bytes = sizeof(UpolsStateHeader) + (size_t)u->ring_slots * u->plane_bytes() + u->zline_bytes();
    return ADSP_OK;
}

int adsp_upols_get_state(adsp_upols* u, void* state, size_t capacity) {
    if (!u || !state) return fail(ADSP_ERR_ARG, "NULL argument");
    size_t need = 0;
    adsp_upols_state_bytes(u, &need);
    if (capacity < need) return fail(ADSP_ERR_ARG, "state buffer of %zu bytes, the engine's state takes %zu (adsp_upols_state_bytes)", capacity, need);
    HIP_TRY(hipSetDevice(u->cfg.device_id));
    HIP_TRY(hipDeviceSynchronize());
    UpolsStateHeader h;
    memset(&h, 0, sizeof h);
    h.magic = kStateMagic;
    h.version = ADSP_ABI_VERSION;
    h.cfg = u->cfg;
    h.ring_slots = u->ring_slots;
    h.R = u->R;
    h.ring_pos = u->ring_pos;
    h.steps_done = u->steps_done;
    h.fwd_done = u->fwd_done;
    h.lfo_copy_len = u->lfo_copy_len;
    h.ring_bytes = (size_t)u->ring_slots * u->plane_bytes();
    h.zline_bytes = u->zline_bytes();
    char* out = static_cast<char*>(state);
    memcpy(out, &h, sizeof h);
    HIP_TRY(hipMemcpy(out + sizeof h, u->ring, h.ring_bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out + sizeof h + h.ring_bytes, u->zline, h.zline_bytes, hipMemcpyDeviceToHost));
    return ADSP_OK;
}

int adsp_upols_set_state(adsp_upols* u, const void* state, size_t bytes) {
    if (!u || !state) return fail(ADSP_ERR_ARG, "NULL argument");
    UpolsStateHeader h;
    if (bytes < sizeof h) return fail(ADSP_ERR_ARG, "state of %zu bytes is shorter than its header", bytes);
    memcpy(&h, state, sizeof h);
    if (h.magic == 0x55504f4cu) return fail(ADSP_ERR_ARG, "the state was taken by an earlier build, whose delay line kept unsplit spectra: it cannot be resumed here");
    if (h.magic != kStateMagic) return fail(ADSP_ERR_ARG, "not a state of a partitioned engine (magic %08x)", h.magic);
    const adsp_upols_config& c = u->cfg;
    if (h.cfg.chunk_size != c.chunk_size || h.cfg.n_channels != c.n_channels || h.cfg.block_size != c.block_size ||
        h.cfg.n_partitions != c.n_partitions || h.cfg.delay != c.delay || h.cfg.sample_format != c.sample_format || h.cfg.max_steps != c.max_steps ||
        h.ring_slots != u->ring_slots || h.R != u->R)
        return fail(ADSP_ERR_ARG, "the state was taken from an engine of another shape (chunk %d, %d channels, block %d, %d partitions, delay %d, format %d, max_steps %d)",
                    h.cfg.chunk_size, h.cfg.n_channels, h.cfg.block_size, h.cfg.n_partitions, h.cfg.delay, h.cfg.sample_format, h.cfg.max_steps);
    if (h.ring_bytes != (size_t)u->ring_slots * u->plane_bytes() || h.zline_bytes != u->zline_bytes() || bytes < sizeof h + h.ring_bytes + h.zline_bytes)
        return fail(ADSP_ERR_ARG, "truncated state (%zu bytes)", bytes);
    if (h.ring_pos < 0 || h.ring_pos >= u->ring_slots || h.steps_done < 0 || h.fwd_done < -1) return fail(ADSP_ERR_ARG, "corrupt state header");
    HIP_TRY(hipSetDevice(c.device_id));
    HIP_TRY(hipDeviceSynchronize());
    const char* in = static_cast<const char*>(state);
    HIP_TRY(hipMemcpy(u->ring, in + sizeof h, h.ring_bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(u->zline, in + sizeof h + h.ring_bytes, h.zline_bytes, hipMemcpyHostToDevice));
    u->ring_pos = h.ring_pos;
    u->steps_done = h.steps_done;
    u->fwd_done = h.fwd_done;
    u->carry_valid = false;  // (not part of a state: the first call after a resume computes its straddling block itself - same values)
    if (u->lfo_len > 0 && h.lfo_copy_len >= 1 && h.lfo_copy_len <= (long long)u->lfo_len + c.chunk_size) u->lfo_copy_len = h.lfo_copy_len;
    return ADSP_OK;
}

// ---- the filter of a running engine: set / get / broadcast (the path's ONE collective, for kernels longer than a transform) ----
int adsp_upols_set_spectra(adsp_upols* u, const float* spectra) {
    if (!u || !spectra) return fail(ADSP_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(u->cfg.device_id));
    HIP_TRY(hipDeviceSynchronize());  // launches in flight still read the old tables (set-up path)
    return upols_upload_tables(u, spectra);
}

int adsp_upols_get_spectra(const adsp_upols* u, float* spectra, size_t n_floats) {
    if (!u || !spectra) return fail(ADSP_ERR_ARG, "NULL argument");
    if (n_floats != upols_spectra_floats(u)) return fail(ADSP_ERR_ARG, "n_floats %zu != n_partitions x 2 (block + 1) = %zu", n_floats, upols_spectra_floats(u));
    memcpy(spectra, u->spectra->data(), n_floats * sizeof(float));
    return ADSP_OK;
}

namespace {
constexpr int kUpolsHdr = 16;
int upols_bcast_buffer(adsp_upols* u) {
    if (!u->d_spectra) HIP_TRY(hipMalloc(&u->d_spectra, (kUpolsHdr + upols_spectra_floats(u)) * sizeof(float)));
    return ADSP_OK;
}
void upols_header(const adsp_upols* u, float (&h)[kUpolsHdr]) {
    const adsp_upols_config& c = u->cfg;
    const float v[kUpolsHdr] = {(float)c.chunk_size, (float)c.block_size, (float)c.n_partitions, (float)c.delay, (float)c.sample_format, 0.f, 0.f, 0.f,
                                0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    memcpy(h, v, sizeof v);
}
// after a collective: the buffer's header must describe THIS engine (a spectrum only means something with the partitioning it was
// designed for); then the tables are rebuilt from what arrived
int upols_adopt(adsp_upols* u, int who) {
    const size_t n = upols_spectra_floats(u);
    std::vector<float> host(kUpolsHdr + n);
    HIP_TRY(hipMemcpy(host.data(), u->d_spectra, host.size() * sizeof(float), hipMemcpyDeviceToHost));
    float mine[kUpolsHdr];
    upols_header(u, mine);
    for (int i = 0; i < 5; ++i)
        if (host[i] != mine[i])
            return fail(ADSP_ERR_ARG, "engine %d: partitioning (chunk %d, block %d, %d partitions, delay %d, format %d) differs from the root's (%d, %d, %d, %d, %d); "
                        "this engine keeps its own filter", who, (int)mine[0], (int)mine[1], (int)mine[2], (int)mine[3], (int)mine[4], (int)host[0], (int)host[1],
                        (int)host[2], (int)host[3], (int)host[4]);
    HIP_TRY(hipDeviceSynchronize());
    return upols_upload_tables(u, host.data() + kUpolsHdr);
}
int upols_stage_root(adsp_upols* u) {
    float h[kUpolsHdr];
    upols_header(u, h);
    HIP_TRY(hipMemcpy(u->d_spectra, h, sizeof h, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(u->d_spectra + kUpolsHdr, u->spectra->data(), upols_spectra_floats(u) * sizeof(float), hipMemcpyHostToDevice));
    return ADSP_OK;
}
}  // namespace

// ONE process, one engine per GPU (ncclCommInitAll inside libadsp): every engine takes over engines[root]'s filter
int adsp_upols_bcast_spectra(adsp_upols* const* engines, int n, int root) {
    if (!engines || n < 1) return fail(ADSP_ERR_ARG, "need at least one engine");
    if (root < 0 || root >= n) return fail(ADSP_ERR_ARG, "root %d out of range 0..%d", root, n - 1);
    for (int i = 0; i < n; ++i) {
        if (!engines[i]) return fail(ADSP_ERR_ARG, "engine %d is NULL", i);
        for (int j = 0; j < i; ++j)
            if (engines[j] == engines[i] || engines[j]->cfg.device_id == engines[i]->cfg.device_id)
                return fail(ADSP_ERR_ARG, "engines %d and %d share device %d: one engine per GPU (RCCL ranks are devices)", j, i, engines[i]->cfg.device_id);
        if (upols_spectra_floats(engines[i]) != upols_spectra_floats(engines[root]))
            return fail(ADSP_ERR_ARG, "engine %d is partitioned differently from the root engine", i);
    }
    std::vector<float*> bufs(n);
    std::vector<int> devs(n);
    std::vector<hipStream_t> streams(n, nullptr);
    for (int i = 0; i < n; ++i) {
        HIP_TRY(hipSetDevice(engines[i]->cfg.device_id));
        int rc = upols_bcast_buffer(engines[i]);
        if (rc) return rc;
        HIP_TRY(hipDeviceSynchronize());
        bufs[i] = engines[i]->d_spectra;
        devs[i] = engines[i]->cfg.device_id;
    }
    HIP_TRY(hipSetDevice(engines[root]->cfg.device_id));
    int rc = upols_stage_root(engines[root]);
    if (rc) return rc;
    if ((rc = adsp::rccl_broadcast(bufs.data(), devs.data(), streams.data(), n, kUpolsHdr + upols_spectra_floats(engines[root]), root))) return rc;
    for (int i = 0; i < n; ++i) {
        HIP_TRY(hipSetDevice(engines[i]->cfg.device_id));
        HIP_TRY(hipDeviceSynchronize());
        if ((rc = upols_adopt(engines[i], i))) return rc;
    }
    return ADSP_OK;
}

// One process per GPU (ncclCommInitRank, the id from adsp_rccl_unique_id on rank 0).  Every rank enters the one collective whatever it
// thinks of its own engine: sizes are fixed by the caller's (chunk, block, partitions), a mismatch shows in the header afterwards.
int adsp_upols_bcast_spectra_rank(adsp_upols* u, const char* unique_id, int rank, int world, int root) {
    if (!u || !unique_id) return fail(ADSP_ERR_ARG, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world || root < 0 || root >= world)
        return fail(ADSP_ERR_ARG, "rank %d / root %d out of range for a world of %d", rank, root, world);
    HIP_TRY(hipSetDevice(u->cfg.device_id));
    int rc = upols_bcast_buffer(u);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (rank == root && (rc = upols_stage_root(u))) return rc;
    if ((rc = adsp::rccl_broadcast_rank(unique_id, rank, world, root, u->cfg.device_id, u->d_spectra, kUpolsHdr + upols_spectra_floats(u), nullptr))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return upols_adopt(u, rank);
}

int adsp_upols_apply_host(adsp_upols* u, const void* in, void* out, int n_steps) {
    if (!u || !in || !out) return fail(ADSP_ERR_ARG, "NULL argument");
    if (n_steps <= 0) return fail(ADSP_ERR_ARG, "n_steps must be positive");
    HIP_TRY(hipSetDevice(u->cfg.device_id));
    const size_t bytes = (size_t)n_steps * u->plane_bytes();
    if (bytes <= adsp::kHostWindowMax && !getenv("ADSP_UPOLS_HOST_STAGED")) {
        // a chunk of one or two channels (Example4's own call through numpy arrays): the launches read / write pinned host memory (capi_common.hpp)
        adsp::HostWindow* w = adsp::host_window(u->cfg.device_id);
        if (!w) return ADSP_ERR_ARG;
        std::lock_guard<std::mutex> lock(w->mu);
        int rc = adsp::host_window_reserve(*w, bytes, bytes);
        if (rc) return rc;
        memcpy(w->in, in, bytes);
        if ((rc = adsp_upols_apply_device(u, w->d_in, w->d_out, n_steps, nullptr))) return rc;
        if ((rc = adsp::host_window_wait(*w, nullptr))) return rc;
        memcpy(out, w->out, bytes);
        return ADSP_OK;
    }
    if (bytes > u->stage_bytes) {
        HIP_TRY(hipDeviceSynchronize());
        if (u->stage_in) (void)hipFree(u->stage_in);
        if (u->stage_out) (void)hipFree(u->stage_out);
        u->stage_in = u->stage_out = nullptr;
        u->stage_bytes = 0;
        HIP_TRY(hipMalloc(&u->stage_in, bytes));
        HIP_TRY(hipMalloc(&u->stage_out, bytes));
        u->stage_bytes = bytes;
    }
    HIP_TRY(hipMemcpy(u->stage_in, in, bytes, hipMemcpyHostToDevice));
    const int rc = adsp_upols_apply_device(u, u->stage_in, u->stage_out, n_steps, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, u->stage_out, bytes, hipMemcpyDeviceToHost));
    return ADSP_OK;
}

}  // extern "C"
