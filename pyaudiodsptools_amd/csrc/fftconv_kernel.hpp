// fftconv_kernel.hpp - fused overlap-save FIR kernel for gfx950 (MI355X), hand-written HIP.
//
// One launch = for every (channel, time block): load F = 2M real samples -> M-point complex
// Stockham FFT in registers + LDS -> real-FFT split, multiply by the filter spectrum, re-pack
// (all three fused in registers) -> M-point inverse FFT -> store the kept samples.
//
// What it replaces in the reference: the per-call numpy pipeline of
// pyAudioDspTools/EffectFFTFilter.py:67-75 (concatenate -> fft(3N) -> * -> ifft(3N) -> slice ->
// astype) and EffectEQ3BandFFT.py:175-211, for ONE channel per call there, for a [channels x
// blocks] grid here.  The reference's 3N complex transforms are not reproduced: the kept slice is
// a plain linear convolution, so a 2N real transform packed as N complex
// points is exact and 3x cheaper.
//
// Design notes (CDNA4, all measured on MI355X - see DESIGN.md section 3/5 and micro/):
//  * wave64; every thread owns P (16, 32 or 64) complex points in VGPRs, element index tid + T*m.
//    Every Stockham pass reads elements  j + q*M/R  (= the thread's own registers) and writes runs
//    of S contiguous elements to LDS, so reads are always bank-conflict-free (64 consecutive
//    8-byte elements per wave) and only the S=1 pass needs an XOR swizzle on the write side.
//  * ds_read_b64/ds_write_b64 on interleaved (re,im) pairs; one LDS buffer of M*8 bytes per
//    transform (32 KiB at N = 4096).
//  * the real-FFT split needs Z[k] and Z[M-k] together.  In-register plans give the last forward pass
//    (radix P/2 with two butterflies per thread - or P/4 with four, .. - Plan::NBL) butterflies j and M/R - j, so both
//    partners are produced in the same thread; the "XL" plan (M = 4096: P = 16, 256 threads, three radix-16 passes) keeps one
//    butterfly per thread and exchanges half of the registers between lanes l and l^32 with
//    v_permlane32_swap.  Either way split + spectrum multiply + re-pack is ONE 2x2 complex matrix per bin
//    pair (pair_op, 16 multiply-adds) and needs no LDS exchange.
//  * inverse FFT = forward FFT on (im, re)-swapped registers: one set of butterflies, one sign.
//  * radix-16 pass twiddles are two-level: 6 loaded, 9 formed in registers (w^(4a+b) = w^(4a) w^b); radix-32 passes
//    likewise (10 loaded, 21 formed).
//  * the large transforms (M = 8192, 16384) run their LDS exchanges in two rounds over HALF a buffer (Plan::HALF), which
//    is what lets a third / second workgroup onto the CU (one M = 16384 transform used to own the CU's LDS and registers).
//  * global accesses are 16 bytes per lane (8 for int16 PCM) with a DPP swap between neighbouring lanes -
//    a dwordx2 costs the TA exactly what a dwordx4 does; chunk selection (ring history vs. the new batch)
//    is resolved once per block into <= F/N + 1 pointers; output stores are non-temporal.
//  * no packed f32 math (half rate on gfx950), no MFMA (no contraction): ~100 flop/sample against
//    8-10 B/sample; sustained multi-step launches run at 1.9-2.0 GHz (power controller), single-step launches at
//    2.5-2.6 GHz (the chip idles between them) - DESIGN.md section 5.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>

// ADSP_ABLATE: tuning-only bitmask that removes one ingredient of the kernel to see what it costs
// (results are then wrong).  1: pass twiddles not loaded  2: pair tables not loaded  4: no LDS exchange
// 8: no global input loads  16: no output stores  32: butterflies replaced by copies  64: I/O aliased onto 8 channels.  Never set in product builds.
#ifndef ADSP_ABLATE
#define ADSP_ABLATE 0
#endif

// ADSP_NT: bit 0 = non-temporal output stores, bit 1 = non-temporal input loads in multi-step launches
#ifndef ADSP_NT
#define ADSP_NT 3
#endif

// ADSP_WIDE_IO: 1 = 16-byte global accesses + DPP lane-pair exchange, 0 = 8-byte accesses (tuning A/B)
#ifndef ADSP_WIDE_IO
#define ADSP_WIDE_IO 1
#endif

#ifndef ADSP_MIN_WAVES
#define ADSP_MIN_WAVES 1
#endif

namespace adsp {

struct KernelArgs {
    const void* ring;      // [ring_slots][C][N] input history ring   (float32 or int16 samples, see S16)
    const void* in;        // [n_steps][C][N] new input (may point into the ring)
    void* out;             // [n_steps][C][N]
    const float4* tw;      // pass twiddles as (w_odd, w_even) pairs: forward passes 1.., then inverse passes 1..
    const float4* pair;    // [R/2][3][T] float4: (wc,g1)_r, (g2_r,wc_r+1), (g1,g2)_r+1 for threads 1..T-1
    const float2* pair0;   // [R+1][3]   thread 0's self-paired butterflies
    const void* zeros;     // >= 4N zero bytes (stands in for chunks that do not exist)
    int ring_pos;          // slot holding the most recent history chunk (time step -1)
    int ring_slots;
    int C;                 // channels
    int n_steps;           // new chunks per channel in `in`
    int V;                 // outputs kept per transform
    int nblk;              // transforms per channel in this launch
    int lookback;          // window start = block output start - lookback
    int j0;                // circular index of the first kept sample
    int ncg;               // channel groups = ceil(C / channels-per-workgroup)
    int N;                 // chunk size (generic-geometry kernel only; the specialised kernels know it at compile time)
    int nh;                // history chunks (generic-geometry kernel only)
    float inv_n;           // 1 / N
    int real_spec;         // the spectrum is real (zero-phase kernel): `pair` holds 3 real constants per bin pair
    int epi_op;            // fused output epilogue (0 = none), see apply_epilogue
    float epi_p0, epi_p1, epi_p2;
    int epi_phase;         // tremolo: LFO table index of this launch's output sample 0
    int epi_replay;        // tremolo: the table index restarts at epi_phase with EVERY chunk (the reference's stuck buffer)
    int win_pairs;         // register PAIRS of the window that are loaded (P/2 = all); the rest is taken as zero: window
                           // positions >= out_offset + V + (kernel taps at negative circular indices) only feed discarded
                           // outputs - a single-step launch of a zero-phase cut filter needs 1.5 N of its 2 N window
    int accumulate;        // 1: add the kept samples to what `out` holds (partitioned FIRs, mixing); 2: and clip the sum
                           // to [-1, 1] (MixSignals).  Plain kernels: generic geometry + mode 1 only; EPI kernels: all.
    // resident ring launches (adsp_apply_ring_resident): the new chunks are ring slots too (`in` is unused) and block b may
    // only start once the producer has PUBLISHED step b - the 32-bit sequence word has reached seq_base + b + 1
    int in_ring;                   // chunks q >= 0 are ring slots (ring_pos + 1 + q) mod ring_slots
    int step_tile;                 // 1 or 4: steps per tile of the step-major workgroup order
    const unsigned* seq;           // device sequence word the producer side bumps after filling a slot (nullptr: no waiting)
    unsigned seq_base;             // value of the word when every step before this launch had been published
    unsigned* seq_fail;            // set to 1 by a workgroup that gave up waiting (its block's outputs are then not written)
    unsigned long long seq_timeout;  // in ticks of the constant 100 MHz clock (wall_clock64)
};

// ------------------------------------------------------------------------------------------
// compile-time plan: M complex points, P points per thread, up to 4 forward radices.
// Inverse radices are the forward ones reversed; the last forward radix must be P/2.
// ------------------------------------------------------------------------------------------
// XL_ = cross-lane pairing: every pass has ONE butterfly per thread (last radix == P) and the real-FFT
// partner of lane l lives in lane l ^ 32 of the same wave (needs T % 64 == 0).
// HALF_ = every LDS exchange runs in two rounds over HALF the buffer (elements < M/2, then the rest): 4 M bytes of LDS per
// transform instead of 8 M, two more barriers per exchange - for the large transforms, whose LDS footprint is what keeps
// a second or third workgroup off the CU.  MINW_ = waves per SIMD the register allocation must leave room for.
template <int M_, int P_, int NP_, int A0, int A1, int A2, int A3, bool XL_ = false, bool HALF_ = false, int MINW_ = ADSP_MIN_WAVES,
          int MINW_AUX_ = MINW_>
struct Plan {
    static constexpr int M = M_, P = P_, NP = NP_, T = M_ / P_;
    static constexpr bool XL = XL_, HALF = HALF_;
    static constexpr int MINW = MINW_;
    // ... and what its int16 / fused-effect / generic-geometry siblings must leave room for (they need more registers)
    static constexpr int minw(bool plain) { return plain ? MINW_ : MINW_AUX_; }
    static constexpr int LDS_ELEMS = HALF_ ? M_ / 2 : M_;
    static constexpr int fwd(int p) { return p == 0 ? A0 : p == 1 ? A1 : p == 2 ? A2 : A3; }
    static constexpr int inv(int p) { return fwd(NP_ - 1 - p); }
    static constexpr int rad(bool inverse, int p) { return inverse ? inv(p) : fwd(p); }
    static constexpr int stride(bool inverse, int p) {  // S = product of the radices before pass p
        int s = 1;
        for (int i = 0; i < p; ++i) s *= rad(inverse, i);
        return s;
    }
    // Twiddles are stored as float4 (one 16-byte load costs the TA exactly what an 8-byte one does).
    //  radix 16: TWO-LEVEL, 3 float4 per j_lo = (w^1,w^2), (w^3,w^4), (w^8,w^12); the other nine powers
    //            w^(4a+b) = w^(4a) * w^b are formed in registers (36 flops instead of 72 bytes of table per butterfly)
    //  radix 32: TWO-LEVEL, 5 float4 per j_lo = (w^1,w^2), (w^3,w^4), (w^5,w^6), (w^7,w^8), (w^16,w^24); the other 21
    //            powers w^(8a+b) = w^(8a) * w^b in registers (5 instead of 16 table loads per butterfly: the large
    //            transforms spend most of their vector-memory instructions on twiddles)
    //  other radices: (w_{2h+1}, w_{2h+2}), h < R/2.
#ifndef ADSP_TW_PREFETCH
#define ADSP_TW_PREFETCH 1
#endif
#ifndef ADSP_LOAD_FENCE
#define ADSP_LOAD_FENCE 1
#endif
#ifndef ADSP_TW2_MIN_S
#define ADSP_TW2_MIN_S 2
#endif
#ifndef ADSP_TW2_RADIX32
#define ADSP_TW2_RADIX32 1
#endif
    static constexpr bool tw_two_level(int radix, int s) {
        return (radix == 16 || (ADSP_TW2_RADIX32 && radix == 32)) && s >= ADSP_TW2_MIN_S;
    }
    static constexpr int tw_rows2(int radix, int s) { return tw_two_level(radix, s) ? (radix == 16 ? 3 : 5) : radix / 2; }
    static constexpr int tw_count(bool inverse) {
        int n = 0;
        for (int p = 1; p < NP_; ++p) n += tw_rows2(rad(inverse, p), stride(inverse, p)) * stride(inverse, p);
        return n;
    }
    static constexpr int tw_offset(bool inverse, int p) {
        int n = inverse ? tw_count(false) : 0;
        for (int i = 1; i < p; ++i) n += tw_rows2(rad(inverse, i), stride(inverse, i)) * stride(inverse, i);
        return n;
    }
    static constexpr int tw_total = tw_count(false) + tw_count(true);
    static constexpr int RL = fwd(NP_ - 1);  // radix of the paired passes (last forward = first inverse)
    static constexpr int NBL = P_ / RL;      // butterflies per thread there: 1 (XL, partner in lane ^ 32) or an even number -
                                             // NBL/2 pairs (j, M/RL - j) whose real-FFT partners meet in registers
    static_assert(XL_ ? NBL == 1 : (NBL >= 2 && NBL % 2 == 0), "last forward radix: P for XL plans, P/2, P/4, .. otherwise");
    // butterfly i of the paired pass of thread tid (in-register plans).  Pair u = i/2: a-side u*T + tid, b-side its mirror
    // NBL*T - (u*T + tid); thread 0's pair 0 holds the two self-paired butterflies 0 and NBL*T/2.
    static __device__ __forceinline__ int paired_bfly(int i, int tid) {
        const int u = i >> 1;
        if ((i & 1) == 0) return u * T + tid;
        return (u == 0 && tid == 0) ? NBL * T / 2 : (NBL - u) * T - tid;
    }
    static_assert(!XL_ || (M_ / P_) % 64 == 0, "XL plans need whole waves");
    static_assert(stride(false, NP_) == M_, "radices must multiply to M");
    static_assert(!HALF_ || M_ >= 2048, "half-buffer exchanges: the swizzle must stay below bit log2(M) - 1");
};

// ------------------------------------------------------------------------------------------
// small DFTs with compile-time twiddles (forward sign, natural order in and out)
// ------------------------------------------------------------------------------------------
__device__ constexpr float kCos32[9] = {1.0f,
                                        0.98078528040323044913f,
                                        0.92387953251128675613f,
                                        0.83146961230254523708f,
                                        0.70710678118654752440f,
                                        0.55557023301960222474f,
                                        0.38268343236508977173f,
                                        0.19509032201612826785f,
                                        0.0f};

// cos(2*pi*i/32) for 0 <= i <= 16
__device__ constexpr float cos32(int i) { return i <= 8 ? kCos32[i < 0 ? 0 : i] : -kCos32[i > 16 ? 0 : 16 - i]; }

// t = W32^IDX * o,  W32 = exp(-2*pi*i/32),  0 <= IDX < 16
template <int IDX>
__device__ __forceinline__ void twmul32(float o_r, float o_i, float& t_r, float& t_i) {
    static_assert(IDX >= 0 && IDX < 16, "");
    if constexpr (IDX == 0) {
        t_r = o_r;
        t_i = o_i;
    } else if constexpr (IDX == 8) {  // -i
        t_r = o_i;
        t_i = -o_r;
    } else if constexpr (IDX == 4) {  // (1-i)/sqrt2
        t_r = (o_r + o_i) * kCos32[4];
        t_i = (o_i - o_r) * kCos32[4];
    } else if constexpr (IDX == 12) {  // (-1-i)/sqrt2
        t_r = (o_i - o_r) * kCos32[4];
        t_i = -(o_r + o_i) * kCos32[4];
    } else {
        constexpr float wr = cos32(IDX);
        constexpr float wi = -cos32(IDX <= 8 ? 8 - IDX : IDX - 8);  // -sin(2 pi IDX / 32)
        t_r = o_r * wr - o_i * wi;
        t_i = o_r * wi + o_i * wr;
    }
}

template <int R>
struct Dft;

template <>
struct Dft<1> {
    static __device__ __forceinline__ void run(const float (&xr)[1], const float (&xi)[1], float (&yr)[1],
                                               float (&yi)[1]) {
        yr[0] = xr[0];
        yi[0] = xi[0];
    }
};

template <>
struct Dft<2> {
    static __device__ __forceinline__ void run(const float (&xr)[2], const float (&xi)[2], float (&yr)[2],
                                               float (&yi)[2]) {
        yr[0] = xr[0] + xr[1];
        yi[0] = xi[0] + xi[1];
        yr[1] = xr[0] - xr[1];
        yi[1] = xi[0] - xi[1];
    }
};

template <int R>
struct Dft {
    static constexpr int H = R / 2;
    template <int K>
    static __device__ __forceinline__ void combine(const float (&Er)[H], const float (&Ei)[H], const float (&Or)[H],
                                                   const float (&Oi)[H], float (&yr)[R], float (&yi)[R]) {
        float tr, ti;
        twmul32<K*(32 / R)>(Or[K], Oi[K], tr, ti);
        yr[K] = Er[K] + tr;
        yi[K] = Ei[K] + ti;
        yr[K + H] = Er[K] - tr;
        yi[K + H] = Ei[K] - ti;
    }
    template <int... K>
    static __device__ __forceinline__ void combine_all(std::integer_sequence<int, K...>, const float (&Er)[H],
                                                       const float (&Ei)[H], const float (&Or)[H],
                                                       const float (&Oi)[H], float (&yr)[R], float (&yi)[R]) {
        (combine<K>(Er, Ei, Or, Oi, yr, yi), ...);
    }
    static __device__ __forceinline__ void run(const float (&xr)[R], const float (&xi)[R], float (&yr)[R],
                                               float (&yi)[R]) {
        float er[H], ei[H], odr[H], odi[H], Er[H], Ei[H], Or[H], Oi[H];
#pragma unroll
        for (int q = 0; q < H; ++q) {
            er[q] = xr[2 * q];
            ei[q] = xi[2 * q];
            odr[q] = xr[2 * q + 1];
            odi[q] = xi[2 * q + 1];
        }
        Dft<H>::run(er, ei, Er, Ei);
        Dft<H>::run(odr, odi, Or, Oi);
        combine_all(std::make_integer_sequence<int, H>{}, Er, Ei, Or, Oi, yr, yi);
    }
};

// DFT-3 (forward sign): W3 = -1/2 - i sqrt(3)/2
__device__ __forceinline__ void dft3(float x0r, float x0i, float x1r, float x1i, float x2r, float x2i, float& y0r, float& y0i,
                                     float& y1r, float& y1i, float& y2r, float& y2i) {
    constexpr float kS = 0.86602540378443864676f;
    const float tr = x1r + x2r, ti = x1i + x2i;
    const float mr = x0r - 0.5f * tr, mi = x0i - 0.5f * ti;
    const float sr = (x1r - x2r) * kS, si = (x1i - x2i) * kS;
    y0r = x0r + tr;
    y0i = x0i + ti;
    y1r = mr + si;  // m - i s
    y1i = mi - sr;
    y2r = mr - si;  // m + i s
    y2i = mi + sr;
}

// DFT-12 as a prime-factor (Good-Thomas) 3 x 4: input n = (4 n1 + 3 n2) mod 12, output k = (4 k1 + 9 k2) mod 12 - then
// W12^(nk) = W3^(n1 k1) W4^(n2 k2), no twiddles between the two stages (96 real additions/multiplications).  The radix of
// the 3 * 2^k-point transforms (F = 1.5 N windows: the minimal window of the reference's cut filters, EffectFFTFilter.py:22-25).
template <>
struct Dft<12> {
    static __device__ __forceinline__ void run(const float (&xr)[12], const float (&xi)[12], float (&yr)[12], float (&yi)[12]) {
        float ar[3][4], ai[3][4];  // [k1][n2]
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) {
            const int i0 = (3 * n2) % 12, i1 = (4 + 3 * n2) % 12, i2 = (8 + 3 * n2) % 12;
            dft3(xr[i0], xi[i0], xr[i1], xi[i1], xr[i2], xi[i2], ar[0][n2], ai[0][n2], ar[1][n2], ai[1][n2], ar[2][n2], ai[2][n2]);
        }
#pragma unroll
        for (int k1 = 0; k1 < 3; ++k1) {
            float br[4], bi[4];
            Dft<4>::run(ar[k1], ai[k1], br, bi);
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                yr[(4 * k1 + 9 * k2) % 12] = br[k2];
                yi[(4 * k1 + 9 * k2) % 12] = bi[k2];
            }
        }
    }
};

// j mod S for a compile-time stride (a mask when S is a power of two; the 3 * 2^k plans have S = 12, 192 in their inverse passes)
template <int S>
__device__ __forceinline__ int mod_stride(int j) {
    if constexpr ((S & (S - 1)) == 0)
        return j & (S - 1);
    else
        return j % S;
}

// ------------------------------------------------------------------------------------------
// LDS addressing.  The exchange written by a pass with S == 1 (runs of R contiguous elements per
// thread) is XOR-swizzled so that 16 consecutive butterflies hit 16 different 8-byte bank pairs;
// later passes write runs of S >= 8 contiguous elements and need nothing.
// ------------------------------------------------------------------------------------------
template <int R, bool SWZ>
__device__ __forceinline__ int lds_phys(int a) {
#if ADSP_ABLATE & 1024
    a = a < 3999 ? a : 3999;  // tuning only (wrong results): what would 5 workgroups per CU buy at M = 4096?
#endif
    if constexpr (!SWZ) {
        return a;
    } else {
        // R = 12 (a = 12 j + r): lanes j, j+4, j+8, j+12 of a 16-lane store group share a & 15 and sit 3 sixteen-element blocks
        // apart: XORing the two low bits with the block index separates them (tools/emulate_lds.py checks every pass)
        constexpr int mask = R == 12 ? 3 : (R < 16 ? R : 16) - 1;
        constexpr int sh = R <= 16 ? 4 : 5;
        return a ^ ((a >> sh) & mask);
    }
}

// ------------------------------------------------------------------------------------------
// one Stockham pass: twiddle, DFT-R on the thread's NB = P/R butterflies, in place in registers
// ------------------------------------------------------------------------------------------
template <class PL, bool INV, int p>
struct Pass {
    static constexpr int P = PL::P, T = PL::T, M = PL::M;
    static constexpr int R = PL::rad(INV, p);
    static constexpr int S = PL::stride(INV, p);
    static constexpr int NB = P / R;
    static constexpr bool PAIRED = INV ? (p == 0) : (p == PL::NP - 1);
    static constexpr bool LAST = (p == PL::NP - 1);
    static constexpr int TWOFF = PL::tw_offset(INV, p);
    static_assert(!PAIRED || NB == PL::NBL, "paired pass: NBL butterflies per thread");

    static __device__ __forceinline__ int bfly(int i, int tid, int ja, int jb) {
        if constexpr (PAIRED) {
            if constexpr (PL::XL) return ja;
            if constexpr (PL::NBL == 2) return i == 0 ? ja : jb;  // (= paired_bfly, already in registers)
            return PL::paired_bfly(i, tid);
        }
        return tid + T * i;
    }

    // two-level twiddle rows of a one-butterfly pass, loadable long before the pass runs (they depend on tid only)
    // ... of one-butterfly passes, and of the two-butterfly passes of the 32-points-per-thread plans (M = 8192: +3 %; at 16
    // points per thread the extra 24 registers buy nothing, at 64 they spill: -5 %)
    static constexpr bool PREFETCHABLE =
        ADSP_TW_PREFETCH && S > 1 && PL::tw_two_level(R, S) && (NB == 1 || (NB == 2 && PL::P == 32));
    static constexpr int TWROWS = PL::tw_rows2(R, S);  // table rows of a two-level pass (3 for radix 16, 5 for radix 32)
    struct Tw3 {
        float4 t[NB < 1 ? 1 : NB][5];
    };
    static __device__ __forceinline__ Tw3 prefetch(const float4* __restrict__ tw, int tid, int ja, int jb) {
        Tw3 r;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int jlo = mod_stride<S>(bfly(i, tid, ja, jb));
#pragma unroll
            for (int k = 0; k < 5; ++k) r.t[i][k] = k < TWROWS ? tw[TWOFF + k * S + jlo] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return r;
    }

    static __device__ __forceinline__ void compute(float (&ar)[P], float (&ai)[P], const float4* __restrict__ tw,
                                                   int tid, int ja, int jb, const Tw3* pre = nullptr) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            float ur[R], ui[R], vr[R], vi[R];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                ur[q] = ar[i + q * NB];
                ui[q] = ai[i + q * NB];
            }
            if constexpr (S > 1 && PL::tw_two_level(R, S)) {
                // two-level twiddles: w^(Ba+b) = w^(Ba) * w^b with B = 4 (radix 16) or 8 (radix 32)
                const int jlo = mod_stride<S>(bfly(i, tid, ja, jb));
                float4 t[5];
#if ADSP_ABLATE & 1
#pragma unroll
                for (int k = 0; k < TWROWS; ++k) t[k] = tw[TWOFF + 2 * k + (jlo & 1)];
#else
                if constexpr (PREFETCHABLE) {
#pragma unroll
                    for (int k = 0; k < TWROWS; ++k) t[k] = pre->t[i][k];
                } else {
#pragma unroll
                    for (int k = 0; k < TWROWS; ++k) t[k] = tw[TWOFF + k * S + jlo];
                }
#endif
                float wr[R], wi[R];
                constexpr int B = R == 16 ? 4 : 8;
                if constexpr (R == 16) {
                    wr[1] = t[0].x; wi[1] = t[0].y; wr[2] = t[0].z; wi[2] = t[0].w;
                    wr[3] = t[1].x; wi[3] = t[1].y; wr[4] = t[1].z; wi[4] = t[1].w;
                    wr[8] = t[2].x; wi[8] = t[2].y; wr[12] = t[2].z; wi[12] = t[2].w;
                } else {
                    wr[1] = t[0].x; wi[1] = t[0].y; wr[2] = t[0].z; wi[2] = t[0].w;
                    wr[3] = t[1].x; wi[3] = t[1].y; wr[4] = t[1].z; wi[4] = t[1].w;
                    wr[5] = t[2].x; wi[5] = t[2].y; wr[6] = t[2].z; wi[6] = t[2].w;
                    wr[7] = t[3].x; wi[7] = t[3].y; wr[8] = t[3].z; wi[8] = t[3].w;
                    wr[16] = t[4].x; wi[16] = t[4].y; wr[24] = t[4].z; wi[24] = t[4].w;
                }
                // every power is used the moment it exists (forming all R of them first keeps 2R registers alive)
                auto rot = [&](int q, float cr, float ci) {
                    const float xr = ur[q], xi = ui[q];
                    ur[q] = xr * cr - xi * ci;
                    ui[q] = xr * ci + xi * cr;
                };
#pragma unroll
                for (int b = 1; b < B; ++b) rot(b, wr[b], wi[b]);
#pragma unroll
                for (int a4 = B; a4 < R; a4 += B) {
                    rot(a4, wr[a4], wi[a4]);
#pragma unroll
                    for (int b = 1; b < B; ++b)
                        rot(a4 + b, wr[a4] * wr[b] - wi[a4] * wi[b], wr[a4] * wi[b] + wi[a4] * wr[b]);
                }
            } else if constexpr (S > 1) {
                const int jlo = mod_stride<S>(bfly(i, tid, ja, jb));
#pragma unroll
                for (int h = 0; h < R / 2; ++h) {
#if ADSP_ABLATE & 1
                    const float4 w = tw[TWOFF + (jlo & 1)];
#else
                    const float4 w = tw[TWOFF + h * S + jlo];
#endif
                    {
                        const float xr = ur[2 * h + 1], xi = ui[2 * h + 1];
                        ur[2 * h + 1] = xr * w.x - xi * w.y;
                        ui[2 * h + 1] = xr * w.y + xi * w.x;
                    }
                    if (2 * h + 2 < R) {
                        const float xr = ur[2 * h + 2], xi = ui[2 * h + 2];
                        ur[2 * h + 2] = xr * w.z - xi * w.w;
                        ui[2 * h + 2] = xr * w.w + xi * w.z;
                    }
                }
            }
#if ADSP_ABLATE & 32
#pragma unroll
            for (int r = 0; r < R; ++r) { vr[r] = ur[r]; vi[r] = ui[r]; }
#else
            Dft<R>::run(ur, ui, vr, vi);
#endif
#pragma unroll
            for (int r = 0; r < R; ++r) {
                ar[i + r * NB] = vr[r];
                ai[i + r * NB] = vi[r];
            }
        }
    }

    // scatter the pass outputs: element (j_hi*R + r)*S + j_lo
    static __device__ __forceinline__ void write(const float (&ar)[P], const float (&ai)[P], float2* lds, int tid,
                                                 int ja, int jb) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int j = bfly(i, tid, ja, jb);
            const int jlo = mod_stride<S>(j);
            const int base = (j - jlo) * R + jlo;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                lds[lds_phys<R, S == 1>(base + r * S)] = make_float2(ar[i + r * NB], ai[i + r * NB]);
            }
        }
    }

    // half-buffer rounds (Plan::HALF): round H moves the elements whose index has top bit H through lds[0 .. M/2).
    // A butterfly's outputs all share that bit (it is the top bit of j_hi), so a round is whole butterflies.
    template <int H>
    static __device__ __forceinline__ void write_half(const float (&ar)[P], const float (&ai)[P], float2* lds, int tid,
                                                      int ja, int jb) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int j = bfly(i, tid, ja, jb);
            const int jlo = mod_stride<S>(j);
            const int base = (j - jlo) * R + jlo;
            if ((base >= M / 2) == (H == 1)) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    lds[lds_phys<R, S == 1>(base + r * S - H * (M / 2))] = make_float2(ar[i + r * NB], ai[i + r * NB]);
            }
        }
    }
    template <int H>
    static __device__ __forceinline__ void read_half(float (&ar)[P], float (&ai)[P], const float2* lds, int tid, int ja,
                                                     int jb) {
        constexpr bool NEXT_PAIRED = !INV && (p + 1 == PL::NP - 1);
        if constexpr (!NEXT_PAIRED) {
#pragma unroll
            for (int m = H * (P / 2); m < (H + 1) * (P / 2); ++m) {  // element tid + T*m >= M/2  <=>  m >= P/2
                const float2 v = lds[lds_phys<R, S == 1>(tid + T * m - H * (M / 2))];
                ar[m] = v.x;
                ai[m] = v.y;
            }
        } else {
            constexpr int Rn = PL::RL, NBn = P / Rn;
            using NXT = Pass<PL, INV, p + 1>;
#pragma unroll
            for (int q = H * (Rn / 2); q < (H + 1) * (Rn / 2); ++q) {  // butterfly indices < M/Rn: the top bit is q's
#pragma unroll
                for (int i = 0; i < NBn; ++i) {
                    const float2 v = lds[lds_phys<R, S == 1>(NXT::bfly(i, tid, ja, jb) + q * (M / Rn) - H * (M / 2))];
                    ar[NBn * q + i] = v.x;
                    ai[NBn * q + i] = v.y;
                }
            }
        }
    }

    // gather what the NEXT pass needs.  Normally element tid + T*m -> register m; when the next
    // pass is the paired one, butterflies ja/jb: element j + q*(M/Rn) -> register i + 2q.
    static __device__ __forceinline__ void read(float (&ar)[P], float (&ai)[P], const float2* lds, int tid, int ja,
                                                int jb) {
        constexpr bool NEXT_PAIRED = !INV && (p + 1 == PL::NP - 1);
        if constexpr (!NEXT_PAIRED) {
#pragma unroll
            for (int m = 0; m < P; ++m) {
                const float2 v = lds[lds_phys<R, S == 1>(tid + T * m)];
                ar[m] = v.x;
                ai[m] = v.y;
            }
        } else {
            constexpr int Rn = PL::RL, NBn = P / Rn;
            using NXT = Pass<PL, INV, p + 1>;
#pragma unroll
            for (int q = 0; q < Rn; ++q) {
#pragma unroll
                for (int i = 0; i < NBn; ++i) {
                    const float2 v = lds[lds_phys<R, S == 1>(NXT::bfly(i, tid, ja, jb) + q * (M / Rn))];
                    ar[NBn * q + i] = v.x;
                    ai[NBn * q + i] = v.y;
                }
            }
        }
    }
};

template <class PL, bool INV, int p>
__device__ __forceinline__ void run_passes(float (&ar)[PL::P], float (&ai)[PL::P], float2* lds,
                                           const float4* __restrict__ tw, int tid, int ja, int jb,
                                           const typename Pass<PL, INV, p>::Tw3* pre = nullptr) {
    using PS = Pass<PL, INV, p>;
    PS::compute(ar, ai, tw, tid, ja, jb, pre);
    if constexpr (!PS::LAST) {
        using NX = Pass<PL, INV, p + 1>;
        // the next pass's twiddles are requested BEFORE the exchange, so their L2 latency hides behind the LDS
        // writes, the barrier and the LDS reads instead of stalling the pass (+3 % measured)
        typename NX::Tw3 nx;
        if constexpr (NX::PREFETCHABLE) nx = NX::prefetch(tw, tid, ja, jb);
#if !(ADSP_ABLATE & 4)
        if constexpr (PL::HALF) {
            // round 0 lands in temporaries: the thread's own outputs for round 1 still sit in ar/ai (a butterfly writes
            // in ONE round, so registers of both halves stay live until the second write); P/2 extra registers, between
            // passes, where pressure is lowest
            float tr[PL::P], ti[PL::P];
            if constexpr (INV || p > 0) __syncthreads();  // everyone is done reading the previous exchange
            PS::template write_half<0>(ar, ai, lds, tid, ja, jb);
            __syncthreads();
            PS::template read_half<0>(tr, ti, lds, tid, ja, jb);
            __syncthreads();
            PS::template write_half<1>(ar, ai, lds, tid, ja, jb);
            __syncthreads();
            PS::template read_half<1>(ar, ai, lds, tid, ja, jb);
#pragma unroll
            for (int m = 0; m < PL::P / 2; ++m) {  // round 0 filled registers 0 .. P/2-1 (both read layouts)
                ar[m] = tr[m];
                ai[m] = ti[m];
            }
        } else {
            if constexpr (INV || p > 0) __syncthreads();  // everyone is done reading the previous exchange
            PS::write(ar, ai, lds, tid, ja, jb);
            __syncthreads();
            PS::read(ar, ai, lds, tid, ja, jb);
        }
#endif
        run_passes<PL, INV, p + 1>(ar, ai, lds, tw, tid, ja, jb, NX::PREFETCHABLE ? &nx : nullptr);
    }
}

// ------------------------------------------------------------------------------------------
// real-FFT split  +  spectrum multiply  +  re-pack for the inverse, on one (k, M-k) pair.
//   in : za = Z[k], zb = Z[M-k]        out: za = Zy[k]/M, zb = Zy[M-k]/M
// With U = Za + conj(Zb), D = Za - conj(Zb), wc = -i W_2M^k, g1 = H[k]/4M, g2 = conj(H[M-k])/4M the textbook chain
//   X[k] ~ U + wc D,  conj(X[M-k]) ~ U - wc D,  P = g1 X1,  Q = g2 X2,  E = P + Q,  O = conj(wc)(P - Q),
//   Zy[k] = E + O,  Zy[M-k] = conj(E - O)
// is linear in (Za, conj Zb) and, because |wc| = 1, collapses to a 2x2 complex matrix with THREE distinct entries
// (s = g1 + g2, d = g1 - g2):   c1 = 2s + 2d Re(wc),  c2 = -2i d Im(wc),  c4 = 2s - 2d Re(wc)
//   Zy[k]   = c1 Za + c2 conj(Zb)
//   Zy[M-k] = conj(c4 conj(Zb) - c2 Za)
// 16 multiply-adds per pair (instead of 32 flops), 24 bytes of table per pair; the host builds c1, c2, c4 in float64.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void pair_op(float& zar, float& zai, float& zbr, float& zbi, const float2 c1,
                                        const float2 c2, const float2 c4) {
    const float o1r = c1.x * zar - c1.y * zai + c2.x * zbr + c2.y * zbi;
    const float o1i = c1.x * zai + c1.y * zar + c2.y * zbr - c2.x * zbi;
    const float tr = c4.x * zbr + c4.y * zbi - c2.x * zar + c2.y * zai;
    const float ti = c4.y * zbr - c4.x * zbi - c2.x * zai - c2.y * zar;
    zar = o1r;
    zai = o1i;
    zbr = tr;
    zbi = -ti;
}

// The same matrix for a REAL spectrum (a symmetric kernel centred on circular index 0, e.g. the reference's low/high
// cut filters): s and d are real, so c1 = (a, 0), c4 = (b, 0), c2 = (0, e) - 8 multiply-adds and 12 bytes per pair.
__device__ __forceinline__ void pair_op_real(float& zar, float& zai, float& zbr, float& zbi, float a, float b, float e) {
    const float o1r = a * zar + e * zbi;
    const float o1i = a * zai + e * zbr;
    const float tr = b * zbr + e * zai;
    const float ti = b * zbi + e * zar;
    zar = o1r;
    zai = o1i;
    zbr = tr;
    zbi = ti;
}

// real-spectrum table: [g][3][T] float4 = (a, b, e) of pairs 4g .. 4g+3 of thread t
__device__ __forceinline__ void load_real_group(const float4* __restrict__ pair, int g, int T, int t, float (&c)[12]) {
    const float4 f0 = pair[(g * 3 + 0) * T + t], f1 = pair[(g * 3 + 1) * T + t], f2 = pair[(g * 3 + 2) * T + t];
    c[0] = f0.x; c[1] = f0.y; c[2] = f0.z; c[3] = f0.w;
    c[4] = f1.x; c[5] = f1.y; c[6] = f1.z; c[7] = f1.w;
    c[8] = f2.x; c[9] = f2.y; c[10] = f2.z; c[11] = f2.w;
}

template <class PL>
__device__ __forceinline__ void spectrum_stage(float (&xr)[PL::P], float (&xi)[PL::P],
                                               const float4* __restrict__ pair, const float2* __restrict__ pair0,
                                               int tid, bool real_spec) {
    constexpr int R = PL::RL, T = PL::T, NB = PL::NBL;
    static_assert(R % 4 == 0, "real-spectrum table packs four pairs per group");
    // registers: pair u of butterflies (a: j = u*T + tid, b: its mirror): a's output r -> [NB*r + 2u], b's -> [NB*r + 2u + 1];
    // bin k = j + (M/R)*r of a meets M - k = output R-1-r of b.  Tables: [u][...][T], the layouts of the NB = 2 case per pair.
#pragma unroll
    for (int u = 0; u < NB / 2; ++u) {
        // Three SEQUENTIAL ifs, not if / else-if / else: the self-paired test is lane-divergent, and LLVM's register
        // liveness runs over the linearised control flow - with an else branch the inputs of the later branch stay live
        // through the earlier one next to its outputs, which doubles the pressure of a stage that rewrites every
        // register (the 64-point plan: 256 VGPRs + 120 B of scratch per lane with else branches, 230 VGPRs and none
        // without).  Each region below rewrites the registers in place for the lanes it runs on.
        const bool self_paired = u == 0 && tid == 0;  // thread 0's pair 0: butterflies 0 and M/R/2 pair within themselves
        if (!self_paired && real_spec) {  // wave-uniform flag
            const float4* tab = pair + u * (R / 4) * 3 * T;
#pragma unroll
            for (int g = 0; g < R / 4; ++g) {
                float c[12];
                load_real_group(tab, g, T, tid, c);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 4 * g + q;
                    pair_op_real(xr[NB * r + 2 * u], xi[NB * r + 2 * u], xr[NB * (R - 1 - r) + 2 * u + 1],
                                 xi[NB * (R - 1 - r) + 2 * u + 1], c[3 * q], c[3 * q + 1], c[3 * q + 2]);
                }
            }
        }
        if (!self_paired && !real_spec) {
            // two bin pairs share three 16-byte table loads
            const float4* tab = pair + u * (R / 2) * 3 * T;
#pragma unroll
            for (int h = 0; h < R / 2; ++h) {
#if ADSP_ABLATE & 2
                const float4 f0 = tab[(tid & 1)], f1 = tab[T + (tid & 1)], f2 = tab[2 * T + (tid & 1)];
#else
                const float4 f0 = tab[(h * 3 + 0) * T + tid];
                const float4 f1 = tab[(h * 3 + 1) * T + tid];
                const float4 f2 = tab[(h * 3 + 2) * T + tid];
#endif
                const int r0 = 2 * h, r1 = 2 * h + 1;
                pair_op(xr[NB * r0 + 2 * u], xi[NB * r0 + 2 * u], xr[NB * (R - 1 - r0) + 2 * u + 1],
                        xi[NB * (R - 1 - r0) + 2 * u + 1], make_float2(f0.x, f0.y), make_float2(f0.z, f0.w),
                        make_float2(f1.x, f1.y));
                pair_op(xr[NB * r1 + 2 * u], xi[NB * r1 + 2 * u], xr[NB * (R - 1 - r1) + 2 * u + 1],
                        xi[NB * (R - 1 - r1) + 2 * u + 1], make_float2(f1.z, f1.w), make_float2(f2.x, f2.y),
                        make_float2(f2.z, f2.w));
            }
        }
        if (self_paired) {
            // thread 0, pair 0 (u == 0 here): the self-paired butterflies j = 0 (registers NB*r) and j = M/R/2 (NB*r + 1).
            // entry 0: k = 0 (DC + Nyquist), entry 1: k = M/2, entries 2..: a-pairs r = 1..R/2-1
            // (k = D r with D (R-r), D = M/R), then b-pairs r = 0..R/2-1 (k = D/2 + D r with D/2 + D (R-1-r)).
            {
                float tr = xr[0], ti = xi[0];
                pair_op(xr[0], xi[0], tr, ti, pair0[0], pair0[1], pair0[2]);
            }
            {
                float tr = xr[NB * (R / 2)], ti = xi[NB * (R / 2)];
                pair_op(xr[NB * (R / 2)], xi[NB * (R / 2)], tr, ti, pair0[3], pair0[4], pair0[5]);
            }
#pragma unroll
            for (int r = 1; r < R / 2; ++r) {
                const int e = 2 + (r - 1);
                pair_op(xr[NB * r], xi[NB * r], xr[NB * (R - r)], xi[NB * (R - r)], pair0[e * 3], pair0[e * 3 + 1],
                        pair0[e * 3 + 2]);
            }
#pragma unroll
            for (int r = 0; r < R / 2; ++r) {
                const int e = 2 + (R / 2 - 1) + r;
                pair_op(xr[NB * r + 1], xi[NB * r + 1], xr[NB * (R - 1 - r) + 1], xi[NB * (R - 1 - r) + 1], pair0[e * 3],
                        pair0[e * 3 + 1], pair0[e * 3 + 2]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// XL plans: the same stage when Z[k] (this lane, register r, k = jx + D r) and Z[M-k] (lane ^ 32, register
// R-1-r) sit in different lanes.  Registers R/2..R-1 are exchanged between the two half-waves with
// v_permlane32_swap (A.upper <-> B.lower; two swaps per register pair (i, i+1) leave partner's old register
// i^1 in my register i), each lane then owns R/2 complete pairs, and the same swaps bring the results home.
// Lanes 0 and 32 of wave 0 hold the two self-paired butterflies (j = 0 and j = D/2) and pair in-lane.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void half_swap(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

template <int P>
__device__ __forceinline__ void exchange_upper_half(float (&xr)[P], float (&xi)[P]) {
#pragma unroll
    for (int i = P / 2; i < P; i += 2) {
        half_swap(xr[i], xr[i + 1]);
        half_swap(xr[i + 1], xr[i]);
        half_swap(xi[i], xi[i + 1]);
        half_swap(xi[i + 1], xi[i]);
    }
}

template <class PL>
__device__ __forceinline__ void spectrum_stage_xl(float (&xr)[PL::P], float (&xi)[PL::P],
                                                  const float4* __restrict__ pair, const float2* __restrict__ pair0,
                                                  int t, bool real_spec) {
    constexpr int R = PL::P, T = PL::T;
    static_assert(R % 8 == 0, "real-spectrum table packs four pairs per group");
    if (t != 0 && t != 32 && real_spec) {  // wave-uniform flag
        exchange_upper_half<R>(xr, xi);
#pragma unroll
        for (int g = 0; g < R / 8; ++g) {
            float c[12];
            load_real_group(pair, g, T, t, c);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 4 * g + q;
                pair_op_real(xr[r], xi[r], xr[(R - 1 - r) ^ 1], xi[(R - 1 - r) ^ 1], c[3 * q], c[3 * q + 1], c[3 * q + 2]);
            }
        }
        exchange_upper_half<R>(xr, xi);
    } else if (t != 0 && t != 32) {
        exchange_upper_half<R>(xr, xi);
#pragma unroll
        for (int h = 0; h < R / 4; ++h) {
#if ADSP_ABLATE & 2
            const float4 f0 = pair[(t & 1)], f1 = pair[T + (t & 1)], f2 = pair[2 * T + (t & 1)];
#else
            const float4 f0 = pair[(h * 3 + 0) * T + t];
            const float4 f1 = pair[(h * 3 + 1) * T + t];
            const float4 f2 = pair[(h * 3 + 2) * T + t];
#endif
            constexpr int dummy = 0;
            (void)dummy;
            const int r0 = 2 * h, r1 = 2 * h + 1;
            pair_op(xr[r0], xi[r0], xr[(R - 1 - r0) ^ 1], xi[(R - 1 - r0) ^ 1], make_float2(f0.x, f0.y),
                    make_float2(f0.z, f0.w), make_float2(f1.x, f1.y));
            pair_op(xr[r1], xi[r1], xr[(R - 1 - r1) ^ 1], xi[(R - 1 - r1) ^ 1], make_float2(f1.z, f1.w),
                    make_float2(f2.x, f2.y), make_float2(f2.z, f2.w));
        }
        exchange_upper_half<R>(xr, xi);
    }
#if !(ADSP_ABLATE & 128)
    else if (t == 0) {
        // j = 0: bins D*r.  entry 0: k = 0, entry 1: k = M/2 (r = R/2), entries 2..: (r, R-r), r = 1..R/2-1
        {
            float tr = xr[0], ti = xi[0];
            pair_op(xr[0], xi[0], tr, ti, pair0[0], pair0[1], pair0[2]);
        }
        {
            float tr = xr[R / 2], ti = xi[R / 2];
            pair_op(xr[R / 2], xi[R / 2], tr, ti, pair0[3], pair0[4], pair0[5]);
        }
#pragma unroll
        for (int r = 1; r < R / 2; ++r) {
            const int e = 2 + (r - 1);
            pair_op(xr[r], xi[r], xr[R - r], xi[R - r], pair0[e * 3], pair0[e * 3 + 1], pair0[e * 3 + 2]);
        }
    } else {
        // j = D/2: bins D/2 + D*r pair (r, R-1-r)
#pragma unroll
        for (int r = 0; r < R / 2; ++r) {
            const int e = 2 + (R / 2 - 1) + r;
            pair_op(xr[r], xi[r], xr[R - 1 - r], xi[R - 1 - r], pair0[e * 3], pair0[e * 3 + 1], pair0[e * 3 + 2]);
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------
// window load / kept-sample store with compile-time chunk geometry.
// FQ = 4 F/N = the transform length in QUARTER chunks (8: F = 2N, 16: F = 4N, 6: F = 1.5 N, the 3 * 2^k plans).
// MPC = 4P/FQ registers per chunk; windows and kept ranges start on quarter-chunk boundaries, so
// (phase RQ in 0..3, register m) -> (chunk index, offset) is known at compile time; a window touches (FQ+3)/4 + 1 chunks.
// ------------------------------------------------------------------------------------------
// neighbour-lane exchange (lane ^ 1) as a DPP quad permute [1,0,3,2]
__device__ __forceinline__ float lane_xor1(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xF, 0xF, true));
}

// 16-byte global accesses.  A thread owns elements tid + T*m (8 bytes each).  For a register pair (2u, 2u+1) the
// even lane of a lane pair fetches 16 bytes at element tid of register 2u (its own value + its neighbour's), the
// odd lane 16 bytes at element tid-1 of register 2u+1 (its neighbour's + its own); one DPP swap per float puts every
// value home.  Same bytes, half the vector-memory instructions (a dwordx2 costs the TA what a dwordx4 does).
// `cb`/`ob` pointers passed in already carry the per-lane adjustment (+2T-2 floats on odd lanes).
template <class PL, int FQ, int RQ>
__device__ __forceinline__ void load_window(const float* const (&cb)[(FQ + 3) / 4 + 1], float (&xr)[PL::P], float (&xi)[PL::P],
                                            bool odd, bool nt, int win_pairs) {
    constexpr int P = PL::P, T = PL::T, MPC = 4 * P / FQ, Q = MPC / 4;
    static_assert(4 * P % FQ == 0 && MPC % 4 == 0, "whole registers per quarter chunk");
    if constexpr (ADSP_WIDE_IO && Q % 2 == 0) {
        // All P/2 loads are issued before the first result is touched: left to itself the scheduler interleaves the
        // lane exchange of the first results with the address arithmetic of the last loads, which then leave one full
        // HBM round trip late.
        // Multi-step launches stream their input once (the overlap of neighbouring blocks is found in L2 either way):
        // non-temporal loads, +2 % measured.  Single-step launches re-read the history ring on the next call: plain loads.
        float4 v[P / 2];
        typedef float v4f __attribute__((ext_vector_type(4)));
        if ((ADSP_NT & 2) && nt) {  // wave-uniform
#pragma unroll
            for (int u = 0; u < P / 2; ++u) {
                const int gi = RQ * Q + 2 * u;  // registers 2u and 2u+1 are always in the same chunk
                const v4f nv = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(cb[gi / MPC] + (gi % MPC) * 2 * T));
                v[u] = make_float4(nv.x, nv.y, nv.z, nv.w);
            }
        } else if (win_pairs >= P / 2) {
#pragma unroll
            for (int u = 0; u < P / 2; ++u) {
                const int gi = RQ * Q + 2 * u;
                const int i = gi / MPC;
                const int off = (gi % MPC) * 2 * T;
#if ADSP_ABLATE & 8
                v[u] = make_float4(static_cast<float>(off + i) * 1e-4f, reinterpret_cast<size_t>(cb[i]) * 1e-20f, 1.f, 2.f);
#else
                v[u] = *reinterpret_cast<const float4*>(cb[i] + off);
#endif
            }
        } else {
            // the tail of the window only feeds discarded outputs (KernelArgs::win_pairs): not fetched
#pragma unroll
            for (int u = 0; u < P / 2; ++u) {
                const int gi = RQ * Q + 2 * u;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (u < win_pairs) v[u] = *reinterpret_cast<const float4*>(cb[gi / MPC] + (gi % MPC) * 2 * T);  // wave-uniform
            }
        }
#if ADSP_LOAD_FENCE
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int u = 0; u < P / 2; ++u) {
            const float sx = lane_xor1(odd ? v[u].x : v[u].z), sy = lane_xor1(odd ? v[u].y : v[u].w);  // what the neighbour needs
            xr[2 * u] = odd ? sx : v[u].x;
            xi[2 * u] = odd ? sy : v[u].y;
            xr[2 * u + 1] = odd ? v[u].z : sx;
            xi[2 * u + 1] = odd ? v[u].w : sy;
        }
    } else {
#pragma unroll
        for (int m = 0; m < P; ++m) {
            const int gi = RQ * Q + m;
            const int i = gi / MPC;
            const int off = (gi % MPC) * 2 * T;
#if ADSP_ABLATE & 8
            const float2 v = make_float2(static_cast<float>(off + i) * 1e-4f, reinterpret_cast<size_t>(cb[i]) * 1e-20f);
#else
            const float2 v = *reinterpret_cast<const float2*>(cb[i] + off);
#endif
            xr[m] = v.x;
            xi[m] = v.y;
        }
    }
}

template <class PL, int FQ, int RQ, bool EPI = false>
__device__ __forceinline__ void store_kept(float* const (&ob)[(FQ + 3) / 4 + 1], const float (&xr)[PL::P],
                                           const float (&xi)[PL::P], int m_lo, int m_hi, bool odd, int mix = 0) {
    constexpr int P = PL::P, T = PL::T, MPC = 4 * P / FQ, Q = MPC / 4;
    if constexpr (ADSP_WIDE_IO && Q % 2 == 0) {
#pragma unroll
        for (int u = 0; u < P / 2; ++u) {
            if (2 * u >= m_lo && 2 * u < m_hi) {  // wave-uniform; kept ranges start/end on even registers
                const int gi = RQ * Q + 2 * u;
                const int i = gi / MPC;
                const int off = (gi % MPC) * 2 * T;
                // even lane stores (own, neighbour's) of register 2u; odd lane (neighbour's, own) of register 2u+1
                const float sx = lane_xor1(odd ? xr[2 * u] : xr[2 * u + 1]);
                const float sy = lane_xor1(odd ? xi[2 * u] : xi[2 * u + 1]);
                float4 v = odd ? make_float4(sx, sy, xr[2 * u + 1], xi[2 * u + 1])
                               : make_float4(xr[2 * u], xi[2 * u], sx, sy);
                if constexpr (EPI) {
                    if (mix) {  // wave-uniform: add what the output already holds (MixSignals: and clip)
                        const float4 old = *reinterpret_cast<const float4*>(ob[i] + off);
                        v = make_float4(v.x + old.x, v.y + old.y, v.z + old.z, v.w + old.w);
                        if (mix == 2)
                            v = make_float4(__builtin_amdgcn_fmed3f(v.x, -1.f, 1.f), __builtin_amdgcn_fmed3f(v.y, -1.f, 1.f),
                                            __builtin_amdgcn_fmed3f(v.z, -1.f, 1.f), __builtin_amdgcn_fmed3f(v.w, -1.f, 1.f));
                    }
                }
#if ADSP_ABLATE & 16
                if (xr[2 * u] == 123.456f) *reinterpret_cast<float4*>(ob[i] + off) = v;
#elif ADSP_NT & 1
                typedef float v4f __attribute__((ext_vector_type(4)));
                v4f nv = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(nv, reinterpret_cast<v4f*>(ob[i] + off));
#else
                *reinterpret_cast<float4*>(ob[i] + off) = v;
#endif
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < P; ++m) {
            if (m >= m_lo && m < m_hi) {  // wave-uniform
                const int gi = RQ * Q + m;
                const int i = gi / MPC;
                const int off = (gi % MPC) * 2 * T;
                float2 v = make_float2(xr[m], xi[m]);
                if constexpr (EPI) {
                    if (mix) {
                        const float2 old = *reinterpret_cast<const float2*>(ob[i] + off);
                        v = make_float2(v.x + old.x, v.y + old.y);
                        if (mix == 2) v = make_float2(__builtin_amdgcn_fmed3f(v.x, -1.f, 1.f), __builtin_amdgcn_fmed3f(v.y, -1.f, 1.f));
                    }
                }
#if ADSP_ABLATE & 16
                if (xr[m] == 123.456f) *reinterpret_cast<float2*>(ob[i] + off) = v;
#else
                *reinterpret_cast<float2*>(ob[i] + off) = v;
#endif
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// int16 PCM samples (the reference's WAV front end, Utility.py:233-238 and :295-312, fused into
// the filter).  An element z[n] = (x[2n], x[2n+1]) is ONE dword; input conversion is (float)int16 and output
// conversion (int16)trunc(y) - the /32768 and *32767 of the reference are folded into the spectrum by the host.
// Same lane-pair trick as above with 8-byte accesses (two elements).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned lane_xor1_u(unsigned v) {
    return __builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ void unpack_s16(unsigned w, float& re, float& im) {
    re = static_cast<float>(static_cast<int>(static_cast<short>(w & 0xffffu)));
    im = static_cast<float>(static_cast<int>(w) >> 16);
}
__device__ __forceinline__ unsigned pack_s16(float re, float im) {
    // (numpy_array * 32767).astype('int16'): truncation toward zero (v_cvt_i32_f32), low 16 bits kept
    const unsigned a = static_cast<unsigned>(static_cast<int>(re)) & 0xffffu;
    const unsigned b = static_cast<unsigned>(static_cast<int>(im)) << 16;
    return a | b;
}

template <class PL, int FQ, int RQ>
__device__ __forceinline__ void load_window_s16(const unsigned* const (&cb)[(FQ + 3) / 4 + 1], float (&xr)[PL::P],
                                                float (&xi)[PL::P], bool odd) {
    constexpr int P = PL::P, T = PL::T, MPC = 4 * P / FQ, Q = MPC / 4;
    if constexpr (ADSP_WIDE_IO && Q % 2 == 0) {
#pragma unroll
        for (int u = 0; u < P / 2; ++u) {
            const int gi = RQ * Q + 2 * u;
            const int i = gi / MPC;
            const int off = (gi % MPC) * T;  // dwords
            const uint2 v = *reinterpret_cast<const uint2*>(cb[i] + off);
            const unsigned sx = lane_xor1_u(odd ? v.x : v.y);
            unpack_s16(odd ? sx : v.x, xr[2 * u], xi[2 * u]);
            unpack_s16(odd ? v.y : sx, xr[2 * u + 1], xi[2 * u + 1]);
        }
    } else {
#pragma unroll
        for (int m = 0; m < P; ++m) {
            const int gi = RQ * Q + m;
            const int i = gi / MPC;
            const int off = (gi % MPC) * T;
            unpack_s16(cb[i][off], xr[m], xi[m]);
        }
    }
}

template <class PL, int FQ, int RQ>
__device__ __forceinline__ void store_kept_s16(unsigned* const (&ob)[(FQ + 3) / 4 + 1], const float (&xr)[PL::P],
                                               const float (&xi)[PL::P], int m_lo, int m_hi, bool odd) {
    constexpr int P = PL::P, T = PL::T, MPC = 4 * P / FQ, Q = MPC / 4;
    if constexpr (ADSP_WIDE_IO && Q % 2 == 0) {
#pragma unroll
        for (int u = 0; u < P / 2; ++u) {
            if (2 * u >= m_lo && 2 * u < m_hi) {
                const int gi = RQ * Q + 2 * u;
                const int i = gi / MPC;
                const int off = (gi % MPC) * T;
                const unsigned w0 = pack_s16(xr[2 * u], xi[2 * u]), w1 = pack_s16(xr[2 * u + 1], xi[2 * u + 1]);
                const unsigned sx = lane_xor1_u(odd ? w0 : w1);
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
                const v2u v = odd ? v2u{sx, w1} : v2u{w0, sx};
                __builtin_nontemporal_store(v, reinterpret_cast<v2u*>(ob[i] + off));
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < P; ++m) {
            if (m >= m_lo && m < m_hi) {
                const int gi = RQ * Q + m;
                const int i = gi / MPC;
                const int off = (gi % MPC) * T;
                ob[i][off] = pack_s16(xr[m], xi[m]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Stateless effects fused on the output registers - zero extra HBM traffic.  Formulas follow the
// reference exactly, quirks included:
//   1 VolumeChange      Utility.py:189-194          y = g x, optionally clipped to [-1, 1]        (p0 = g, p1 = clip flag)
//   2 CreateSoftClipper EffectSoftClipper.py:38-45  y = sgn(x) (1 - |min(|x|,1) - 1|^p0)         (p0 = drive + 1)
//   3 CreateHardDistortion EffectHardDistortion.py:30-41  (0.8 + 0.2 sin((a - 0.8)/0.2)) sgn, with a = |x| if |x| <= 0.8
//                       else sgn(x) (so x < -0.8 lands on sin(-9): the reference's asymmetry is kept)
//   4 CreateSaturator   EffectSaturator.py:41-49    knee above p0, (p0+1)/2 above 1, makeup p1, mode p2 (1 hard, 2 soft)
//   6 CreateBitCrusher  _EffectBitCrusher.py:8-12   int16(trunc(32767 x)) // 512 / 64 (private, unexported in the reference)
//   5 CreateTremolo     EffectTremolo.py:19-47      periodic LFO table of p2 samples: gain = 1 - p0/2 + p0/2 sin(2 pi p1 n)
// ------------------------------------------------------------------------------------------
// One sample through effect OP (compile-time) - the reference's expressions in float32, with hardware log2/exp2/sin/rcp
// (each ~1 ulp; the parity tolerance is 1e-5 of full scale).
template <int OP>
__device__ __forceinline__ float effect_sample(float x, float p0, float p1, float p2) {
    if constexpr (OP == 1) {  // volume: p0 = linear gain, p1 = clip flag
        const float y = p0 * x;
        return p1 != 0.f ? __builtin_amdgcn_fmed3f(y, -1.f, 1.f) : y;
    } else if constexpr (OP == 2) {  // soft clipper: sign(x) (1 - |min(|x|,1) - 1|^p0), p0 = drive + 1
        const float a = fminf(fabsf(x), 1.f);
        const float t = 1.f - __builtin_amdgcn_exp2f(p0 * __builtin_amdgcn_logf(1.f - a));  // log2(0) = -inf -> 0
        return __builtin_copysignf(t, x);
    } else if constexpr (OP == 3) {  // hard distortion (0.8 linear limit; beyond it the SIGN is the amplitude)
        const float sgn = x >= 0.f ? 1.f : -1.f;
        float a = fabsf(x);
        a = a <= 0.8f ? a : sgn;
        // v_sin_f32 takes revolutions: (a - 0.8) / 0.2 rad = (a - 0.8) * 5 / (2 pi) rev, |arg| < 1.5 rev
        const float comp = 0.2f * __builtin_amdgcn_sinf((a - 0.8f) * 0.795774715459f);
        return (0.8f + comp) * sgn;
    } else if constexpr (OP == 4) {  // saturator: p0 = threshold, p1 = linear make-up gain, p2 = 1 hard / 2 soft
        float a = fabsf(x);
        const float u = a - p0;
        float r = u * __builtin_amdgcn_rcpf(1.f - p0);
        r = p2 == 2.f ? r * r : r;
        const float knee = p0 + u * __builtin_amdgcn_rcpf(1.f + r);
        a = a > p0 ? knee : a;
        a = a > 1.f ? (p0 + 1.f) * 0.5f : a;
        return __builtin_copysignf(a, x) * p1;
    } else if constexpr (OP == 6) {  // bit crusher: int16(trunc(32767 x)) floor-divided by 512, over 64
        const int q = static_cast<int>(static_cast<short>(static_cast<int>(x * 32767.f)));  // astype('int16') wraps
        return static_cast<float>(q >> 9) * 0.015625f;
    } else {
        return x;
    }
}

// Tremolo gain of LFO table index n (EffectTremolo.py:20-23): ((sin(2 pi f n / fs) / 2) + 0.5) depth + (1 - depth)
__device__ __forceinline__ float tremolo_gain(int n, float depth, float rev_per_sample) {
    return fmaf(0.5f * depth, __builtin_amdgcn_sinf(static_cast<float>(n) * rev_per_sample), 1.f - 0.5f * depth);
}
// n mod len for 0 <= n < 2^24 (float reciprocal, one correction either way)
__device__ __forceinline__ int small_mod(int n, int len, float inv_len) {
    int r = n - static_cast<int>(static_cast<float>(n) * inv_len) * len;
    r += r < 0 ? len : 0;
    r -= r >= len ? len : 0;
    return r;
}

__device__ __forceinline__ float epilogue_value(float x, int op, float p0, float p1, float p2) {
    switch (op) {
        case 1: return effect_sample<1>(x, p0, p1, p2);
        case 2: return effect_sample<2>(x, p0, p1, p2);
        case 3: return effect_sample<3>(x, p0, p1, p2);
        case 4: return effect_sample<4>(x, p0, p1, p2);
        case 6: return effect_sample<6>(x, p0, p1, p2);
        default: return x;
    }
}

template <int OP, int P>
__device__ __forceinline__ void epilogue_loop(float (&xr)[P], float (&xi)[P], const KernelArgs& a) {
#pragma unroll
    for (int m = 0; m < P; ++m) {
        xr[m] = effect_sample<OP>(xr[m], a.epi_p0, a.epi_p1, a.epi_p2);
        xi[m] = effect_sample<OP>(xi[m], a.epi_p0, a.epi_p1, a.epi_p2);
    }
}

// Tremolo: the reference multiplies the stream by a periodic table of len = p2 samples (EffectTremolo.py:20-47);
// register m of thread tid holds output times tau0 + 2 T m (+1 for the imaginary part), tau0 = block start - j0 + 2 tid.
template <int P, int T>
__device__ __forceinline__ void tremolo_loop(float (&xr)[P], float (&xi)[P], const KernelArgs& a, int tau0) {
    const int len = static_cast<int>(a.epi_p2);
    const float inv_len = 1.f / a.epi_p2;
    if (a.epi_replay) {
        // every chunk replays the table from epi_phase: index = (phase + time within the chunk) mod len
        const float inv_n = 1.f / static_cast<float>(a.N);
        int r0 = tau0 % a.N;  // once per thread; negative for samples that are not kept
        r0 += r0 < 0 ? a.N : 0;
#pragma unroll
        for (int m = 0; m < P; ++m) {
            const int r = small_mod(r0 + 2 * T * m, a.N, inv_n);
            const int r1 = r + 1 == a.N ? 0 : r + 1;
            xr[m] *= tremolo_gain(small_mod(a.epi_phase + r, len, inv_len), a.epi_p0, a.epi_p1);
            xi[m] *= tremolo_gain(small_mod(a.epi_phase + r1, len, inv_len), a.epi_p0, a.epi_p1);
        }
        return;
    }
    int base = (a.epi_phase + tau0) % len;  // once per thread; tau0 may be negative for samples that are not kept
    base += base < 0 ? len : 0;
#pragma unroll
    for (int m = 0; m < P; ++m) {
        const int n = small_mod(base + 2 * T * m, len, inv_len);
        const int n1 = n + 1 == len ? 0 : n + 1;
        xr[m] *= tremolo_gain(n, a.epi_p0, a.epi_p1);
        xi[m] *= tremolo_gain(n1, a.epi_p0, a.epi_p1);
    }
}

template <int P, int T>
__device__ __forceinline__ void apply_epilogue(float (&xr)[P], float (&xi)[P], const KernelArgs& a, int tau0) {
    switch (a.epi_op) {  // wave-uniform
        case 0: return;
        case 5: return tremolo_loop<P, T>(xr, xi, a, tau0);
        case 1: return epilogue_loop<1>(xr, xi, a);
        case 2: return epilogue_loop<2>(xr, xi, a);
        case 3: return epilogue_loop<3>(xr, xi, a);
        case 4: return epilogue_loop<4>(xr, xi, a);
        case 6: return epilogue_loop<6>(xr, xi, a);
    }
}

// ------------------------------------------------------------------------------------------
// the transform core shared by both kernels: forward FFT -> spectrum stage -> inverse FFT, in registers + LDS
// ------------------------------------------------------------------------------------------
template <class PL>
__device__ __forceinline__ void transform_block(float (&xr)[PL::P], float (&xi)[PL::P], float2* lds, const KernelArgs& a,
                                                int tid) {
    constexpr int T = PL::T;
    int ja, jb;
    if constexpr (PL::XL) {
        // lanes 0-31 of wave w: butterflies 32w + l; lanes 32-63: their partners T - (32w + l)
        const int lo = 32 * (tid >> 6) + (tid & 31);
        ja = (tid & 32) ? (tid == 32 ? T / 2 : T - lo) : lo;
        jb = 0;
    } else {
        ja = tid;                                                     // = paired_bfly(0, tid)
        jb = (tid == 0) ? PL::NBL * T / 2 : PL::NBL * T - tid;        // = paired_bfly(1, tid)
    }

    run_passes<PL, false, 0>(xr, xi, lds, a.tw, tid, ja, jb);
#if !(ADSP_ABLATE & 256)
    if constexpr (PL::XL)
        spectrum_stage_xl<PL>(xr, xi, a.pair, a.pair0, tid, a.real_spec != 0);
    else
        spectrum_stage<PL>(xr, xi, a.pair, a.pair0, tid, a.real_spec != 0);
#endif
    run_passes<PL, true, 0>(xi, xr, lds, a.tw, tid, ja, jb);  // inverse = forward on swapped parts
}

// ------------------------------------------------------------------------------------------
// Resident ring launches: block `step` waits until the producer side has published its chunk.  Thread 0 polls the
// sequence word (system-scope loads: the writer is a copy engine or another kernel), gives up after seq_timeout ticks of
// the 100 MHz clock - a consumer launched without a producer must not hang the GPU - and tells the whole workgroup through
// LDS.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool wait_for_step(const KernelArgs& a, unsigned step, unsigned* lds_flag) {
    if (threadIdx.x == 0) {
        const unsigned need = a.seq_base + step + 1u;
        unsigned ok = 1;
        if (static_cast<int>(__hip_atomic_load(a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - need) < 0) {
            const unsigned long long t0 = wall_clock64();
            while (static_cast<int>(__hip_atomic_load(a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - need) < 0) {
                __builtin_amdgcn_s_sleep(16);
                if (wall_clock64() - t0 > a.seq_timeout) {
                    ok = 0;
                    __hip_atomic_store(a.seq_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
        lds_flag[0] = ok;
    }
    __syncthreads();
    const unsigned ok = lds_flag[0];
    __syncthreads();  // the flag word is part of the exchange buffer
    // No cache invalidation is needed, only program order (the loads below may not be hoisted above the poll): a launch
    // covers at most ring_slots - history steps, so every slot it reads is written ONCE, before the first read of it in this
    // launch - no CU's L1 (invalidated at kernel start) and no XCD's L2 can hold an older copy fetched during the launch.
    // An agent-scope acquire here (buffer_inv sc1 in every workgroup) cost 2.7x the whole kernel (profiles/r3_resident.txt).
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return ok != 0;
}

// ------------------------------------------------------------------------------------------
// the kernel: one workgroup = CPB channels x one time block
// ------------------------------------------------------------------------------------------
template <class PL, int CPB, int FQ, bool S16 = false, bool EPI = false>
__global__ __launch_bounds__(PL::T* CPB, PL::minw(!S16 && !EPI)) void fftconv_kernel(const KernelArgs a) {
    constexpr int M = PL::M, P = PL::P, T = PL::T;
    constexpr int N = 8 * M / FQ;  // chunk size (FQ quarter chunks per transform of 2M samples)
    constexpr int NCH = (FQ + 3) / 4 + 1;  // chunks a window can touch
    static_assert((N & (N - 1)) == 0, "specialised kernels: power-of-two chunks");
    constexpr int LOGN = __builtin_ctz(N);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);

    const int tid = static_cast<int>(threadIdx.x) % T;
    const int grp = static_cast<int>(threadIdx.x) / T;
    lds += grp * PL::LDS_ELEMS;

    // blockIdx -> (channel group, time block).  Blocks b % 8 land on XCD b % 8 (observed, speed
    // only): keep one channel group's consecutive time blocks on one XCD so the overlapping part
    // of their windows is served by that XCD's L2.
    const int lin = static_cast<int>(blockIdx.x);
    const int xcd = lin & 7;
    const int idx = lin >> 3;
    int cgl = idx / a.nblk;
    int blk = idx - cgl * a.nblk;
    if (a.seq) {
        // resident ring launches run STEP-major (every channel group of step 0, then step 1, ..): workgroups are dispatched in
        // index order, and one that waits for a later step must not hold a CU slot before those of earlier steps have one.
        // When every step was already published at launch time the order is tiles of 4 steps instead (every channel group of
        // steps 0..3, then 4..7; a channel group's 4 steps are neighbours, so their window overlap is an L2 hit: -6 %)
        const int ST = a.step_tile;  // 1 while steps are still to be published, 4 when the launch found them all published
        const int ncgl = (a.ncg + 7) >> 3;
        const int tile = idx / (ncgl * ST);
        const int rem = idx - tile * (ncgl * ST);
        cgl = rem / ST;
        blk = tile * ST + (rem - cgl * ST);
        if (blk >= a.nblk) return;
    }
    const int cg = cgl * 8 + xcd;
    if (cg >= a.ncg) return;  // whole workgroup leaves together
    const int c = cg * CPB + grp;
    const bool chan_ok = CPB == 1 ? true : (c < a.C);

    if (a.seq) {  // wave-uniform: a resident ring launch (V == N: block b is step b and reads nothing newer than its own chunk)
        if (!wait_for_step(a, static_cast<unsigned>(blk), reinterpret_cast<unsigned*>(smem_raw))) return;
    }
    const int o = blk * a.V;        // first output-time of this block (multiple of N/4)
    const int t0 = o - a.lookback;  // first input-time of the window (multiple of N/4)
    // Sample storage unit U: float (2 per element) or, for int16 PCM, one dword per element.
    using U = typename std::conditional<S16, unsigned, float>::type;
    constexpr int UPE = S16 ? 1 : 2;                                        // units per complex element
    const size_t plane = (static_cast<size_t>(a.C) << LOGN) / 2 * UPE;      // one [C][N] chunk batch, in units
    // wide I/O: odd lanes address the neighbour pair of the NEXT register (element tid-1, one register = T elements on)
    constexpr bool WIDE = ADSP_WIDE_IO && ((4 * P / FQ) / 4) % 2 == 0;
    const bool odd = WIDE && (tid & 1);
    const size_t lane_off = static_cast<size_t>(UPE) * (tid + (odd ? T - 1 : 0));
#if ADSP_ABLATE & 64
    const size_t chan_off = (static_cast<size_t>(c & 7) << LOGN) / 2 * UPE + lane_off;  // tuning: L2-resident I/O
#else
    const size_t chan_off = (static_cast<size_t>(c) << LOGN) / 2 * UPE + lane_off;
#endif

    // The window touches at most NCH chunks.  Resolve each to a pointer once: ring history, new
    // input, or the zero page for chunks that do not exist yet / channels past the end.
    const int q0 = t0 >> LOGN;  // floor: chunk of the window start, < 0 = history
    const U* cb[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int q = q0 + i;
        const U* base = static_cast<const U*>(a.zeros) + lane_off;
        if (chan_ok && q < a.n_steps) {
            if (q < 0 || a.in_ring) {
                int slot = a.ring_pos + 1 + q;
                slot += (slot < 0) ? a.ring_slots : 0;
                slot = slot < 0 ? 0 : slot;  // (older than the history: never dereferenced with data that matters)
                slot -= (slot >= a.ring_slots) ? a.ring_slots : 0;  // resident launches run ahead of ring_pos, at most one lap
                base = static_cast<const U*>(a.ring) + static_cast<size_t>(slot) * plane + chan_off;
            } else {
                base = static_cast<const U*>(a.in) + static_cast<size_t>(q) * plane + chan_off;
            }
        }
        cb[i] = base;
    }

    float xr[P], xi[P];
    if constexpr (S16) {
        switch ((t0 & (N - 1)) >> (LOGN - 2)) {  // window phase within its first chunk, in quarter chunks
            case 0: load_window_s16<PL, FQ, 0>(cb, xr, xi, odd); break;
            case 1: load_window_s16<PL, FQ, 1>(cb, xr, xi, odd); break;
            case 2: load_window_s16<PL, FQ, 2>(cb, xr, xi, odd); break;
            default: load_window_s16<PL, FQ, 3>(cb, xr, xi, odd); break;
        }
    } else {
        switch ((t0 & (N - 1)) >> (LOGN - 2)) {
            case 0: load_window<PL, FQ, 0>(cb, xr, xi, odd, a.n_steps > 1, a.win_pairs); break;
            case 1: load_window<PL, FQ, 1>(cb, xr, xi, odd, a.n_steps > 1, a.win_pairs); break;
            case 2: load_window<PL, FQ, 2>(cb, xr, xi, odd, a.n_steps > 1, a.win_pairs); break;
            default: load_window<PL, FQ, 3>(cb, xr, xi, odd, a.n_steps > 1, a.win_pairs); break;
        }
    }

    transform_block<PL>(xr, xi, lds, a, tid);
    if constexpr (EPI) apply_epilogue<P, T>(xr, xi, a, blk * a.V - a.j0 + 2 * tid);  // separate instantiation: the plain kernel pays nothing

    // kept samples: circular indices [j0, j0 + keep) -> registers m_lo <= m < m_hi; register m holds
    // output-time o - j0 + 2T*m.  s = o - j0 may be negative: split into chunk part and phase.
    const int total = a.n_steps << LOGN;
    const int keep = (total - o) < a.V ? (total - o) : a.V;
    const int m_lo = a.j0 / (2 * T), m_hi = (a.j0 + keep) / (2 * T);
    const int s = o - a.j0;
    const int k0 = s >> LOGN;  // floor
    U* ob[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        int k = k0 + i;
        k = k < 0 ? 0 : (k < a.n_steps ? k : a.n_steps - 1);  // clamped ones are never stored to
        ob[i] = static_cast<U*>(a.out) + static_cast<size_t>(k) * plane + chan_off;
    }
    if (chan_ok) {
        if constexpr (S16) {
            switch ((s & (N - 1)) >> (LOGN - 2)) {
                case 0: store_kept_s16<PL, FQ, 0>(ob, xr, xi, m_lo, m_hi, odd); break;
                case 1: store_kept_s16<PL, FQ, 1>(ob, xr, xi, m_lo, m_hi, odd); break;
                case 2: store_kept_s16<PL, FQ, 2>(ob, xr, xi, m_lo, m_hi, odd); break;
                default: store_kept_s16<PL, FQ, 3>(ob, xr, xi, m_lo, m_hi, odd); break;
            }
        } else {
            switch ((s & (N - 1)) >> (LOGN - 2)) {
                case 0: store_kept<PL, FQ, 0, EPI>(ob, xr, xi, m_lo, m_hi, odd, a.accumulate); break;
                case 1: store_kept<PL, FQ, 1, EPI>(ob, xr, xi, m_lo, m_hi, odd, a.accumulate); break;
                case 2: store_kept<PL, FQ, 2, EPI>(ob, xr, xi, m_lo, m_hi, odd, a.accumulate); break;
                default: store_kept<PL, FQ, 3, EPI>(ob, xr, xi, m_lo, m_hi, odd, a.accumulate); break;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Generic-geometry kernel: ANY chunk size divisible by 4.  Overlap-save does not need the transform
// tied to the chunk: blocks of V kept samples tile a channel's time axis, and every 16-byte access (4 samples, never
// straddling a chunk because N % 4 == 0 and all block geometry is a multiple of 4) finds its chunk with one
// float reciprocal division per lane.  Same transform core, ~15 % more VALU for the address arithmetic.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void locate_chunk(int tau_biased, int N, float inv_n, int& q, int& r) {
    q = static_cast<int>(static_cast<float>(tau_biased) * inv_n);  // estimate: off by one at most while tau < 2^24,
    r = tau_biased - q * N;                                        // by a few chunks beyond (float(tau) is inexact there)
    while (r < 0) {
        --q;
        r += N;
    }
    while (r >= N) {
        ++q;
        r -= N;
    }
}

template <class PL, int CPB, bool S16 = false, bool EPI = false>
__global__ __launch_bounds__(PL::T* CPB, PL::minw(false)) void fftconv_generic_kernel(const KernelArgs a) {
    constexpr int P = PL::P, T = PL::T;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const int tid = static_cast<int>(threadIdx.x) % T;
    const int grp = static_cast<int>(threadIdx.x) / T;
    lds += grp * PL::LDS_ELEMS;

    const int lin = static_cast<int>(blockIdx.x);
    const int xcd = lin & 7;
    const int idx = lin >> 3;
    const int cgl = idx / a.nblk;
    const int blk = idx - cgl * a.nblk;
    const int cg = cgl * 8 + xcd;
    if (cg >= a.ncg) return;
    const int c = cg * CPB + grp;
    const bool chan_ok = c < a.C;

    using U = typename std::conditional<S16, unsigned, float>::type;  // storage unit: float, or a dword of two int16
    constexpr int SPU = S16 ? 2 : 1;                                    // samples per unit
    const int N = a.N;
    const size_t plane = static_cast<size_t>(a.C) * N / SPU;
    const size_t chan_units = static_cast<size_t>(c) * N / SPU;
    const bool odd = tid & 1;
    const int o = blk * a.V;
    const int t0 = o - a.lookback + a.nh * N;  // window start on the biased (>= 0) time axis: history chunk -nh is chunk 0

    float xr[P], xi[P];
    // all loads first, then the lane exchanges (see load_window); non-temporal in multi-step launches
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    typename std::conditional<S16, v2u, v4f>::type raw[P / 2];
    const bool nt = (ADSP_NT & 2) && a.n_steps > 1;
#pragma unroll
    for (int u = 0; u < P / 2; ++u) {
        // even lane: elements (tid, tid+1) of register 2u; odd lane: elements (tid-1, tid) of register 2u+1
        const int elem = (tid - (odd ? 1 : 0)) + T * (2 * u + (odd ? 1 : 0));
        int q, r;
        locate_chunk(t0 + 2 * elem, N, a.inv_n, q, r);
        q -= a.nh;
        const U* base = static_cast<const U*>(a.zeros);
        size_t off = r / SPU;
        if (chan_ok && q < a.n_steps) {
            if (q < 0) {
                int slot = a.ring_pos + 1 + q;
                slot += (slot < 0) ? a.ring_slots : 0;
                base = static_cast<const U*>(a.ring) + static_cast<size_t>(slot) * plane;
            } else {
                base = static_cast<const U*>(a.in) + static_cast<size_t>(q) * plane;
            }
            off += chan_units;
        }
        if constexpr (S16) {
            raw[u] = *reinterpret_cast<const v2u*>(base + off);
        } else {
            raw[u] = nt ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(base + off))
                        : *reinterpret_cast<const v4f*>(base + off);
        }
    }
#if ADSP_LOAD_FENCE
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int u = 0; u < P / 2; ++u) {
        if constexpr (S16) {
            const v2u v = raw[u];
            const unsigned sx = lane_xor1_u(odd ? v.x : v.y);
            unpack_s16(odd ? sx : v.x, xr[2 * u], xi[2 * u]);
            unpack_s16(odd ? v.y : sx, xr[2 * u + 1], xi[2 * u + 1]);
        } else {
            const v4f v = raw[u];
            const float sx = lane_xor1(odd ? v.x : v.z), sy = lane_xor1(odd ? v.y : v.w);
            xr[2 * u] = odd ? sx : v.x;
            xi[2 * u] = odd ? sy : v.y;
            xr[2 * u + 1] = odd ? v.z : sx;
            xi[2 * u + 1] = odd ? v.w : sy;
        }
    }

    transform_block<PL>(xr, xi, lds, a, tid);
    if constexpr (EPI) apply_epilogue<P, T>(xr, xi, a, blk * a.V - a.j0 + 2 * tid);  // separate instantiation: the plain kernel pays nothing

    const long long total_ll = static_cast<long long>(a.n_steps) * N;
    const int total = static_cast<int>(total_ll);
    const int m_lo = a.j0 / (2 * T), m_hi = (a.j0 + a.V) / (2 * T);
#pragma unroll
    for (int u = 0; u < P / 2; ++u) {
        if (2 * u >= m_lo && 2 * u < m_hi) {  // wave-uniform: j0 and V are multiples of 4T (whole register pairs)
            const int elem = (tid - (odd ? 1 : 0)) + T * (2 * u + (odd ? 1 : 0));
            const int tau = o + 2 * elem - a.j0;
            float sx, sy;
            unsigned w0 = 0, w1 = 0, swx = 0;
            if constexpr (S16) {
                w0 = pack_s16(xr[2 * u], xi[2 * u]);
                w1 = pack_s16(xr[2 * u + 1], xi[2 * u + 1]);
                swx = lane_xor1_u(odd ? w0 : w1);
            } else {
                sx = lane_xor1(odd ? xr[2 * u] : xr[2 * u + 1]);
                sy = lane_xor1(odd ? xi[2 * u] : xi[2 * u + 1]);
            }
            if (chan_ok && tau < total) {
                int k, r;
                locate_chunk(tau, N, a.inv_n, k, r);
                U* dst = static_cast<U*>(a.out) + static_cast<size_t>(k) * plane + chan_units + r / SPU;
                if constexpr (S16) {
                    typedef unsigned v2u __attribute__((ext_vector_type(2)));
                    const v2u v = odd ? v2u{swx, w1} : v2u{w0, swx};
                    __builtin_nontemporal_store(v, reinterpret_cast<v2u*>(dst));
                } else {
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    v4f v = odd ? v4f{sx, sy, xr[2 * u + 1], xi[2 * u + 1]} : v4f{xr[2 * u], xi[2 * u], sx, sy};
                    if (a.accumulate) v += *reinterpret_cast<const v4f*>(dst);  // partial sum of an earlier partition / mix bus
                    if constexpr (EPI) {
                        if (a.accumulate == 2)
                            v = v4f{__builtin_amdgcn_fmed3f(v.x, -1.f, 1.f), __builtin_amdgcn_fmed3f(v.y, -1.f, 1.f),
                                    __builtin_amdgcn_fmed3f(v.z, -1.f, 1.f), __builtin_amdgcn_fmed3f(v.w, -1.f, 1.f)};
                    }
                    __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(dst));
                }
            }
        }
    }
}

}  // namespace adsp
