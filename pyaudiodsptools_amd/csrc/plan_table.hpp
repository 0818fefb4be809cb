// plan_table.hpp - the kernel plan registry shared by the per-format translation units and the C ABI.
// One PlanInfo = one instantiation of adsp::fftconv_kernel (transform size, F/N ratio, sample format).
#pragma once
#include <hip/hip_runtime.h>

#include "fftconv_kernel.hpp"

namespace adsp {

struct PlanInfo {
    int M, FQ, P, T, CPB, NP, XL;  // FQ = 4 * fft_size / chunk_size (8: F = 2N, 16: F = 4N, 6: F = 1.5 N); XL = cross-lane pairing plan
    int rad[4];
    int tw_total;
    int lds_bytes;
    hipError_t (*launch)(const KernelArgs&, int grid, hipStream_t);
    hipError_t (*prepare)();
    // the same transform behind the generic-geometry kernel (any chunk size divisible by 4, FQ is ignored)
    hipError_t (*launch_generic)(const KernelArgs&, int grid, hipStream_t);
    hipError_t (*prepare_generic)();
    // ... and for chunk sizes that are not multiples of 4 (dword accesses; float32 plain kernels only, nullptr otherwise)
    hipError_t (*launch_unaligned)(const KernelArgs&, int grid, hipStream_t);
    hipError_t (*prepare_unaligned)();
};

// one persistent kernel of the live sessions: M = N complex points (F = 2N), lookback LQ quarter chunks
struct LivePlanInfo {
    int M, CPB, LQ, T;
    int P, NP, XL, rad[4], tw_total;  // the plan itself: a session whose plan is not the engine's builds its own tables from this
    int tw_in_lds;                    // the kernel keeps the pass twiddles in LDS
    hipError_t (*launch)(const LiveArgs&, int grid, hipStream_t);
    hipError_t (*capacity)(int* blocks_per_cu);  // sets the kernel's LDS attribute, asks the occupancy API
};

#include "plan_table_core.inc"
#ifdef ADSP_WITH_F64
namespace f64 {
#include "plan_table_core.inc"
}  // namespace f64
#endif

// M (complex points) x FQ (= 4 F/N, the transform length in quarter chunks) -> plan.  Radices forward (inverse = reversed); last forward radix is P/2
// (in-register pairing) or P (XL, cross-lane pairing) - see fftconv_kernel.hpp.
// The two large transforms run their LDS exchanges over half a buffer (Plan::HALF) so that more than one or two
// workgroups fit a CU - measured on MI355X (profiles/r2_shapes_session4.txt):
//   M = 8192 : 32 points per thread, 168 VGPRs (3 waves per SIMD), 32 KiB of LDS -> THREE workgroups per CU  (+9 % over two)
//   M = 16384: 64 points per thread, 4 waves per transform, 64 KiB of LDS        -> TWO workgroups per CU    (+13 % over one)
// M = 4096 as the 4N transform of N = 2048 batches (round 5): 32 points per thread in TWO waves, radices 32 x 8 x 16, in-register pairing,
// half-buffer exchange (16 KiB), 128 VGPRs without scratch: EIGHT independent transforms per CU.  Rounds 1 - 4 ran every M = 4096 transform on
// the "XL" plan (16 points per thread in four waves, partner bins in lane ^ 32, 32 KiB: four per CU) - chosen in round 1 against a 32-point
// plan that spilled (190 VGPRs) before the spectrum stage lost its else branches and the exchange its second half.  Measured A/B on one box
// (profiles/r5_m4096_two_wave_plan.txt): N = 2048 batches 543 500 against 500 800 Msamples/s (+8.5 %, 0.549 of the roofline), the EQ at
// N = 2048 +4 ... 8 %, one launch per chunk with 8192 channels 95.9 against 103.7 us of kernel time, resident launches 48.9 against 52.5 us
// per step - but one launch per chunk at config 2's own shape (4096 channels: exactly four generations of XL workgroups) 51.5 against 49.4 us,
// library-pipelined 45.9 either way.  So the F = 2N column (per-chunk launches at N = 4096, the generic kernel's first match) keeps the XL
// plan and the F = 4N column takes this one; each stays selectable for the other column (variants 22 / 26 / 27, plans_var.hip).
// ... and the same recipe one more size down for the F = 4N column of M = 2048 (N = 1024 batches): 32 points per thread in ONE wave, radices
// 32 x 4 x 16, half exchange (8 KiB), 128 VGPRs + 12 B: 550 200 against 515 900 Msamples/s (+6.7 %, 0.556 of the roofline); per chunk at
// N = 2048 it loses 3 % (49.7 against 47.8 us), so the F = 2N column keeps 16 points per thread in two waves (variants 28 - 30).
#ifndef ADSP_PLAN_2048_4N
#define ADSP_PLAN_2048_4N Plan<2048, 32, 3, 32, 4, 16, 1, false, true, 4, 3>
#endif
#ifndef ADSP_PLAN_4096
#define ADSP_PLAN_4096 Plan<4096, 32, 3, 32, 8, 16, 1, false, true, 4, 3>
#endif
#ifndef ADSP_PLAN_8192
#define ADSP_PLAN_8192 Plan<8192, 32, 3, 32, 16, 16, 1, false, true, 4, 3>
#endif
// M = 16384 (round 6): 32 points per thread in 512 threads, radices 32 x 32 x 16, half exchange (64 KiB): two workgroups of eight waves
// per CU at 128 VGPRs.  Rounds 2 - 5 ran 64 points per thread in 256 threads (16 x 32 x 32, 239 VGPRs, two waves per SIMD:
// ADSP_PLAN_16384_64PT), against which this form was +0.5 ... 2 % in round 5; this round's instruction-level rework (fused butterflies,
// DPP selects, hand-split exchange addresses) returned two to three times as much to the 32-point plans, and the A/B now reads
// +8.3 % for the fused chain (config 5) and +7.8 % for the EQ at N = 8192 (profiles/r6_fused_butterflies_ab.txt, session 13a).
// The long-kernel engines keep the 64-point form for their blocks of 16384 (adsp_upols.hip: measured there, the other way round).
#ifndef ADSP_PLAN_16384
#define ADSP_PLAN_16384 Plan<16384, 32, 3, 32, 32, 16, 1, false, true, 4, 3>
#endif
#ifndef ADSP_PLAN_16384_64PT
#define ADSP_PLAN_16384_64PT Plan<16384, 64, 3, 16, 32, 32, 1, false, true, 2>
#endif
// M = 3072 = 3 * 2^10 (round 3): the F = 1.5 N window of single-step launches of the cut filters at N = 4096 - N + 2d samples
// is all their kept chunk needs, a 2N transform does 37 % more butterfly work per kept sample.  48 points per thread,
// ONE wave per transform (radices 16 x 16 x 12, the DFT-12 a twiddle-free 3 x 4 prime-factor butterfly), half-buffer
// exchange (12 KiB per transform).
#ifndef ADSP_PLAN_3072
#define ADSP_PLAN_3072 Plan<3072, 48, 3, 16, 16, 12, 1, false, true, 2>
#endif
#define ADSP_PLAN_LIST(S16, EPI, U4)                                                   \
    make_plan<Plan<64, 16, 2, 8, 8, 1, 1>, 16, 8, S16, EPI, U4>(),                        \
    make_plan<Plan<128, 16, 2, 16, 8, 1, 1>, 8, 8, S16, EPI, U4>(),                       \
    make_plan<Plan<128, 16, 2, 16, 8, 1, 1>, 8, 16, S16, EPI, U4>(),                       \
    make_plan<Plan<256, 16, 3, 4, 8, 8, 1>, 4, 8, S16, EPI, U4>(),                        \
    make_plan<Plan<256, 16, 3, 4, 8, 8, 1>, 4, 16, S16, EPI, U4>(),                        \
    make_plan<Plan<512, 16, 3, 16, 4, 8, 1>, 2, 8, S16, EPI, U4>(),                       \
    make_plan<Plan<512, 16, 3, 16, 4, 8, 1>, 2, 16, S16, EPI, U4>(),                       \
    make_plan<Plan<1024, 16, 3, 16, 8, 8, 1>, 1, 8, S16, EPI, U4>(),                      \
    make_plan<Plan<1024, 16, 3, 16, 8, 8, 1>, 1, 16, S16, EPI, U4>(),                      \
    make_plan<Plan<2048, 16, 3, 16, 16, 8, 1>, 1, 8, S16, EPI, U4>(),                     \
    make_plan<ADSP_PLAN_2048_4N, 1, 16, S16, EPI, U4>(),                                   \
    make_plan<Plan<4096, 16, 3, 16, 16, 16, 1, true>, 1, 8, S16, EPI, U4>(),              \
    make_plan<ADSP_PLAN_4096, 1, 16, S16, EPI, U4>(),                                      \
    make_plan<ADSP_PLAN_8192, 1, 8, S16, EPI, U4>(),                                      \
    make_plan<ADSP_PLAN_8192, 1, 16, S16, EPI, U4>(),                                      \
    make_plan<ADSP_PLAN_16384, 1, 16, S16, EPI, U4>(),                                    \
    make_plan<ADSP_PLAN_3072, 1, 6, S16, EPI, U4>()

// tables live in plans_f32.hip / plans_s16.hip (internal linkage there: host-only data, the device pass only
// needs to see the instantiations)
const PlanInfo* plans_f32(int* count);
const PlanInfo* plans_s16(int* count);
const PlanInfo* plans_f32_epi(int* count);  // same list, kernels with the fused output effect (float32 only)
const PlanInfo* variants_f32(int* count);  // A/B alternatives, ADSP_PLAN_VARIANT=<n>
const LivePlanInfo* live_plans(int* count);  // plans_live.hip
hipError_t live_publish(unsigned* seq, unsigned value, hipStream_t s);
hipError_t live_publish_out(unsigned* seq, unsigned value, unsigned long long* entry, void* d_out, hipStream_t s);  // + the step's output address
const PlanInfo* plans_s16_f64(int* count);  // int16 samples, float64 arithmetic (namespace adsp::f64 kernels): plans_s16_f64.hip

}  // namespace adsp
