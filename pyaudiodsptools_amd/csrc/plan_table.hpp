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
};

template <class PL, int CPB>
constexpr int lds_bytes() {
#if ADSP_ABLATE & 1024
    if (PL::M == 4096) return 32000;
#endif
#if ADSP_ABLATE & 4096
    if (PL::M == 4096) return 44 * 1024;  // tuning: three workgroups per CU instead of four
#endif
    return PL::LDS_ELEMS * CPB * (int)sizeof(float2);
}

template <class PL, int CPB, int FQ, bool S16, bool EPI>
hipError_t launch_impl(const KernelArgs& a, int grid, hipStream_t s) {
    hipLaunchKernelGGL((fftconv_kernel<PL, CPB, FQ, S16, EPI>), dim3(grid), dim3(PL::T * CPB), (lds_bytes<PL, CPB>()), s, a);
    return hipGetLastError();
}

template <class PL, int CPB, int FQ, bool S16, bool EPI>
hipError_t prepare_impl() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&fftconv_kernel<PL, CPB, FQ, S16, EPI>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<PL, CPB>());
}

template <class PL, int CPB, bool S16, bool EPI>
hipError_t launch_generic_impl(const KernelArgs& a, int grid, hipStream_t s) {
    hipLaunchKernelGGL((fftconv_generic_kernel<PL, CPB, S16, EPI>), dim3(grid), dim3(PL::T * CPB), (lds_bytes<PL, CPB>()), s, a);
    return hipGetLastError();
}

template <class PL, int CPB, bool S16, bool EPI>
hipError_t prepare_generic_impl() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&fftconv_generic_kernel<PL, CPB, S16, EPI>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<PL, CPB>());
}

template <class PL, int CPB, int FQ, bool S16, bool EPI>
constexpr PlanInfo make_plan() {
    return PlanInfo{PL::M, FQ, PL::P, PL::T, CPB, PL::NP, PL::XL ? 1 : 0, {PL::fwd(0), PL::fwd(1), PL::fwd(2), PL::fwd(3)},
                    PL::tw_total, lds_bytes<PL, CPB>(), &launch_impl<PL, CPB, FQ, S16, EPI>, &prepare_impl<PL, CPB, FQ, S16, EPI>,
                    &launch_generic_impl<PL, CPB, S16, EPI>, &prepare_generic_impl<PL, CPB, S16, EPI>};
}

// M (complex points) x FQ (= 4 F/N, the transform length in quarter chunks) -> plan.  Radices forward (inverse = reversed); last forward radix is P/2
// (in-register pairing) or P (XL, cross-lane pairing) - see fftconv_kernel.hpp.
// The two large transforms run their LDS exchanges over half a buffer (Plan::HALF) so that more than one or two
// workgroups fit a CU - measured on MI355X (profiles/r2_shapes_session4.txt):
//   M = 8192 : 32 points per thread, 168 VGPRs (3 waves per SIMD), 32 KiB of LDS -> THREE workgroups per CU  (+9 % over two)
//   M = 16384: 64 points per thread, 4 waves per transform, 64 KiB of LDS        -> TWO workgroups per CU    (+13 % over one)
#ifndef ADSP_PLAN_8192
#define ADSP_PLAN_8192 Plan<8192, 32, 3, 32, 16, 16, 1, false, true, 4, 3>
#endif
#ifndef ADSP_PLAN_16384
#define ADSP_PLAN_16384 Plan<16384, 64, 3, 16, 32, 32, 1, false, true, 2>
#endif
// M = 3072 = 3 * 2^10 (round 3): the F = 1.5 N window of single-step launches of the cut filters at N = 4096 - N + 2d samples
// is all their kept chunk needs, a 2N transform does 37 % more butterfly work per kept sample.  48 points per thread,
// ONE wave per transform (radices 16 x 16 x 12, the DFT-12 a twiddle-free 3 x 4 prime-factor butterfly), half-buffer
// exchange (12 KiB per transform).
#ifndef ADSP_PLAN_3072
#define ADSP_PLAN_3072 Plan<3072, 48, 3, 16, 16, 12, 1, false, true, 2>
#endif
#define ADSP_PLAN_LIST(S16, EPI)                                                    \
    make_plan<Plan<64, 16, 2, 8, 8, 1, 1>, 16, 8, S16, EPI>(),                        \
    make_plan<Plan<128, 16, 2, 16, 8, 1, 1>, 8, 8, S16, EPI>(),                       \
    make_plan<Plan<128, 16, 2, 16, 8, 1, 1>, 8, 16, S16, EPI>(),                       \
    make_plan<Plan<256, 16, 3, 4, 8, 8, 1>, 4, 8, S16, EPI>(),                        \
    make_plan<Plan<256, 16, 3, 4, 8, 8, 1>, 4, 16, S16, EPI>(),                        \
    make_plan<Plan<512, 16, 3, 16, 4, 8, 1>, 2, 8, S16, EPI>(),                       \
    make_plan<Plan<512, 16, 3, 16, 4, 8, 1>, 2, 16, S16, EPI>(),                       \
    make_plan<Plan<1024, 16, 3, 16, 8, 8, 1>, 1, 8, S16, EPI>(),                      \
    make_plan<Plan<1024, 16, 3, 16, 8, 8, 1>, 1, 16, S16, EPI>(),                      \
    make_plan<Plan<2048, 16, 3, 16, 16, 8, 1>, 1, 8, S16, EPI>(),                     \
    make_plan<Plan<2048, 16, 3, 16, 16, 8, 1>, 1, 16, S16, EPI>(),                     \
    make_plan<Plan<4096, 16, 3, 16, 16, 16, 1, true>, 1, 8, S16, EPI>(),              \
    make_plan<Plan<4096, 16, 3, 16, 16, 16, 1, true>, 1, 16, S16, EPI>(),              \
    make_plan<ADSP_PLAN_8192, 1, 8, S16, EPI>(),                                      \
    make_plan<ADSP_PLAN_8192, 1, 16, S16, EPI>(),                                      \
    make_plan<ADSP_PLAN_16384, 1, 16, S16, EPI>(),                                    \
    make_plan<ADSP_PLAN_3072, 1, 6, S16, EPI>()

// tables live in plans_f32.hip / plans_s16.hip (internal linkage there: host-only data, the device pass only
// needs to see the instantiations)
const PlanInfo* plans_f32(int* count);
const PlanInfo* plans_s16(int* count);
const PlanInfo* plans_f32_epi(int* count);  // same list, kernels with the fused output effect (float32 only)
const PlanInfo* variants_f32(int* count);  // A/B alternatives, ADSP_PLAN_VARIANT=<n>

}  // namespace adsp
