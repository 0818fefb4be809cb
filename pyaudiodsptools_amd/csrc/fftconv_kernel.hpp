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
//  * round 3: the kernels themselves live in fftconv_core.inc, written against `real` and included here once for float
//    (namespace adsp) and, under ADSP_WITH_F64, once for double (namespace adsp::f64: the exact-FFT int16 engines); the
//    geometry parameter is FQ = 4F/N (8, 16, or 6 for the 3 * 2^k plan); the spectrum stage is three sequential ifs
//    (an if / else chain doubled the register pressure: DESIGN.md section 3.1); resident ring launches wait per step on a
//    sequence word (wait_for_step).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>

// ADSP_ABLATE: tuning-only bitmask that removes one ingredient of the kernel to see what it costs
// (results are then wrong).  1: pass twiddles not loaded  2: pair tables not loaded  4: no LDS exchange
// 8: no global input loads  16: no output stores  32: butterflies replaced by copies  64: I/O aliased onto 8 channels
// 8192: half-buffer exchanges forward 0 -> 1 and last inverse without their workgroup barriers.  16384: twiddle powers not formed.  Never set in product builds.
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

// ADSP_PERSIST: tuning-only (results stay correct): a workgroup of the specialised kernel loops over KernelArgs::blk_iters consecutive blocks
#ifndef ADSP_PERSIST
#define ADSP_PERSIST 0
#endif

// Both switches change what the kernels compute or how they take their arguments: they exist in the tuning library only
// (make tuning -> libadsp_tuning.so, -DADSP_TUNING_BUILD; tools/build_ablations.sh).  A product build that sets one does not compile.
#if (ADSP_ABLATE != 0 || ADSP_PERSIST != 0) && !defined(ADSP_TUNING_BUILD)
#error "ADSP_ABLATE / ADSP_PERSIST are tuning switches (wrong results by construction / another argument convention): build them with `make tuning` (-DADSP_TUNING_BUILD), never into libadsp.so"
#endif

namespace adsp {

struct KernelArgs {
    const void* ring;      // [ring_slots][C][N] input history ring   (float32 or int16 samples, see S16)
    const void* in;        // [n_steps][C][N] new input (may point into the ring)
    void* out;             // [n_steps][C][N]
    const void* tw;        // real4[]: pass twiddles as (w_odd, w_even) pairs: forward passes 1.., then inverse passes 1..
    const void* pair;      // real4 [R/2][3][T]: (wc,g1)_r, (g2_r,wc_r+1), (g1,g2)_r+1 for threads 1..T-1
    const void* pair0;     // real2 [R+1][3]   thread 0's self-paired butterflies
                           // (real = float, or double in the f64 flavour of the kernels: namespace adsp::f64)
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
    int nt_lo, nt_hi;      // multi-step launches: register pairs nt_lo <= u < nt_hi of the window are loaded non-temporally, the others - the
                           // head and the tail, which the neighbouring blocks of the channel read as well - with plain loads that leave
                           // the lines in L2 for them (0, P/2: everything non-temporal)
    int accumulate;        // 1: add the kept samples to what `out` holds (partitioned FIRs, mixing); 2: and clip the sum
                           // to [-1, 1] (MixSignals).  Plain kernels: generic geometry + mode 1 only; EPI kernels: all.
    // resident ring launches (adsp_apply_ring_resident): the new chunks are ring slots too (`in` is unused) and block b may
    // only start once the producer has PUBLISHED step b - the 32-bit sequence word has reached seq_base + b + 1
    int in_ring;                   // chunks q >= 0 are ring slots (ring_pos + 1 + q) mod ring_slots
    int step_tile;                 // steps per tile of the step-major workgroup order (1, or n_steps when all are published)
    const unsigned* seq;           // device sequence word the producer side bumps after filling a slot (nullptr: no waiting)
    unsigned seq_base;             // value of the word when every step before this launch had been published
    unsigned* seq_fail;            // set to 1 by a workgroup that gave up waiting (its block's outputs are then not written)
    unsigned long long seq_timeout;  // in ticks of the constant 100 MHz clock (wall_clock64)
    int blk_iters;                 // tuning builds with -DADSP_PERSIST=1 only (tools/build_ablations.sh persist): consecutive time blocks ONE workgroup
                                   // runs through, so that the store drain of block b overlaps the window loads of block b + 1; 1 everywhere else
    const void* self;              // ... and a copy of these arguments in device memory, which such a workgroup re-reads per block (scalar loads)
                                   // instead of keeping thirty of them in SGPRs across its loop
};

// Live sessions (adsp_live_*, round 4): ONE persistent launch consumes ring steps as they are published, for as long as the
// host wants.  Every workgroup owns its channel group for the whole session and keeps the `lookback` samples of history the
// next window needs IN REGISTERS, so each input sample is read from memory exactly once; the grid is no larger than the GPU
// holds (the host checks), so a workgroup that waits for a publication never blocks the dispatch of another.
struct LiveArgs {
    KernelArgs k;                  // ring, tables, C, ncg, lookback, j0, ring_pos / ring_slots: as for a per-step launch (k.out unused)
    void* out;                     // output ring [out_slots][C][N]: step s of this session goes to slot s % out_slots
    int out_slots;
    unsigned first_pub;            // value of the sequence word when every step before this session had been published
    unsigned max_steps;            // the session ends after this many steps (or when stopped)
    unsigned* seq;                 // device word (fine-grained): steps published so far; device-side producers bump it themselves
    const volatile unsigned* host_seq;  // host-mapped word a HOST producer bumps with a plain store (no HIP call); the relay workgroup
                                   // forwards it into *seq
    unsigned* progress;            // [ncg] device words: steps this workgroup had completed when it left (adsp_live_stop reads them)
    unsigned* arrivals;            // [arrival_slots][16 shards][16 words]: shard cg % 16 of slot s % arrival_slots counts the workgroups of that
                                   // residue that have completed step s - one returnless atomic per workgroup and step, each shard a 64-byte
                                   // line of its own (atomics on ONE word serialise: ~90 per us, and a step of config 3 has 4096 of them);
                                   // after lap L of the slots a shard reads (workgroups of its residue) * (L + 1)
    unsigned arrival_slots;        // a power of two > ring_slots: no workgroup is ever that many steps ahead of the slowest one
    unsigned* done;                // device word: min over progress[] (maintained by the relay workgroup)
    volatile unsigned* host_done;  // the same, host-mapped: the host reads it without a HIP call
    const volatile unsigned* host_stop;  // host-mapped: non-zero = end the session once every published step is consumed
    unsigned* stop;                // device copy of it (relay), polled by waiting workgroups
    unsigned* fail;                // set by a workgroup that gave up waiting (time-out)
    unsigned long long timeout;    // ticks of the 100 MHz clock a workgroup waits for ONE step (0 = for ever)
    int load_mode;                 // 0 plain, 1 non-temporal, 2 system-scope (sc0 sc1) loads of the new chunk (tuning; default 2)
    unsigned long long* trace;     // tuning (ADSP_LIVE_TRACE=<first step>): workgroup trace_wg stamps s_memtime at six points of 64 steps
    unsigned trace_first;
    int trace_wg;
    int relay_mode;                // tuning: 1 = the relay leaves at once (publications must then come from the device side)
    // per-step output pointers (round 5: adsp_apply_ring riding a session, adsp_ring_set_pipeline(engine, 3)): entry s & out_table_mask holds
    // the device address the N outputs per channel of step s go to ([C][N], like one slot of `out`); the publisher writes it BEFORE it
    // bumps the sequence word.  nullptr: the output ring `out` above.
    const unsigned long long* out_table;
    unsigned out_table_mask;
};

#define ADSP_F64 0
#include "fftconv_core.inc"
#undef ADSP_F64

// The same kernels in float64 (adsp::f64::fftconv_kernel ...): the exact-FFT engines of int16 PCM batches.
#ifdef ADSP_WITH_F64
namespace f64 {
#define ADSP_F64 1
#include "fftconv_core.inc"
#undef ADSP_F64
}  // namespace f64
#endif

}  // namespace adsp
