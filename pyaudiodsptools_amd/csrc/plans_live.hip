// plans_live.hip - the persistent kernels of live sessions (adsp_live_*, fftconv_core.inc: fftconv_live_kernel): float32
// samples, F = 2N transforms of the stream geometry, one instantiation per (chunk size, lookback in quarter chunks).
// Lookback 5/4 N = the reference's cut filters (EffectFFTFilter.py), 7/4 N = its 3-band FFT EQ (EffectEQ3BandFFT.py).
#include "plan_table.hpp"

namespace {
using namespace adsp;

// device-side publication of the next step(s): enqueued on the producer's stream behind the commands that filled the slot
__global__ void live_publish_kernel(unsigned* seq, unsigned value) {
    __hip_atomic_fetch_max(seq, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ... and the same with the step's output address (sessions fed by adsp_apply_ring): the table entry first, then the publication with
// release semantics at system scope - a worker that has seen the publication reads the entry behind it
__global__ void live_publish_out_kernel(unsigned* seq, unsigned value, unsigned long long* entry, unsigned long long d_out) {
    __hip_atomic_store(entry, d_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_fetch_max(seq, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// exchange buffers + the pass twiddles, which a session keeps in LDS when they fit beside eight workgroups per CU; the plan with one
// wave per channel (17 workgroups per CU) keeps rows 0 of the tables only and forms the other powers in registers
template <class PL, int CPB>
constexpr bool live_tw_in_lds() {
    return !PL::XL || PL::P >= 16;
}
// ... and, for the 8-points-per-thread plan, the history rows (LQ quarter chunks of N = M samples per channel)
template <class PL, int CPB, int LQ>
constexpr int live_lds_bytes() {
    return lds_bytes<PL, CPB>() + (live_tw_in_lds<PL, CPB>() ? PL::tw_total : PL::tw1_total) * (int)sizeof(real4) +
           (PL::P <= 8 ? CPB * LQ * (PL::M / 4) * (int)sizeof(float) : 0);
}

template <class PL, int CPB, int LQ>
hipError_t live_launch(const LiveArgs& a, int grid, hipStream_t s) {
    hipLaunchKernelGGL((fftconv_live_kernel<PL, CPB, 8, LQ>), dim3(grid), dim3(PL::T * CPB), (live_lds_bytes<PL, CPB, LQ>()), s, a);
    return hipGetLastError();
}

template <class PL, int CPB, int LQ>
hipError_t live_capacity(int* blocks_per_cu) {
    const void* fn = reinterpret_cast<const void*>(&fftconv_live_kernel<PL, CPB, 8, LQ>);
    hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, live_lds_bytes<PL, CPB, LQ>());
    if (err != hipSuccess) return err;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, fn, PL::T * CPB, live_lds_bytes<PL, CPB, LQ>());
}

template <class PL, int CPB, int LQ>
constexpr LivePlanInfo make_live() {
    return LivePlanInfo{PL::M, CPB, LQ, PL::T, PL::P, PL::NP, PL::XL ? 1 : 0, {PL::fwd(0), PL::fwd(1), PL::fwd(2), PL::fwd(3)}, PL::tw_total,
                        live_tw_in_lds<PL, CPB>() ? 1 : 0, &live_launch<PL, CPB, LQ>, &live_capacity<PL, CPB, LQ>};
}

#define ADSP_LIVE_FOR(LQ)                                              \
    make_live<Plan<128, 16, 2, 16, 8, 1, 1>, 8, LQ>(),                 \
    make_live<Plan<256, 16, 3, 4, 8, 8, 1>, 4, LQ>(),                  \
    make_live<Plan<512, 16, 3, 16, 4, 8, 1>, 2, LQ>(),                 \
    make_live<Plan<1024, 16, 3, 16, 8, 8, 1>, 1, LQ>(),                \
    make_live<Plan<2048, 16, 3, 16, 16, 8, 1>, 1, LQ>(),               \
    make_live<Plan<4096, 16, 3, 16, 16, 16, 1, true>, 1, LQ>()

// First match wins.  Config 3's shape (N = 512, the 3-band EQ's lookback) runs ONE WAVE PER CHANNEL: 8 points per thread, three
// radix-8 passes, partner bins in lane ^ 32 - half the dependent work per wave and twice the waves of the 16-point plan, which
// is what a session needs when every channel group is one serial chain of steps (4 waves per SIMD instead of 2).
const LivePlanInfo kLive[] = {make_live<Plan<512, 8, 3, 8, 8, 8, 1, true>, 1, 7>(), ADSP_LIVE_FOR(5), ADSP_LIVE_FOR(7)};
}  // namespace

const adsp::LivePlanInfo* adsp::live_plans(int* count) {
    *count = sizeof(kLive) / sizeof(kLive[0]);
    return kLive;
}

hipError_t adsp::live_publish(unsigned* seq, unsigned value, hipStream_t s) {
    hipLaunchKernelGGL(live_publish_kernel, dim3(1), dim3(1), 0, s, seq, value);
    return hipGetLastError();
}

hipError_t adsp::live_publish_out(unsigned* seq, unsigned value, unsigned long long* entry, void* d_out, hipStream_t s) {
    hipLaunchKernelGGL(live_publish_out_kernel, dim3(1), dim3(1), 0, s, seq, value, entry, reinterpret_cast<unsigned long long>(d_out));
    return hipGetLastError();
}
