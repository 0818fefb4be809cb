// adsp_guard.hip - non-finite inputs the way the reference treats them (include/adsp.h: adsp_nonfinite_guard).
//
// The reference transforms chunks k-2, k-1, k as ONE 3N-point buffer (EffectFFTFilter.py:67-72, EffectEQ3BandFFT.py:175-179): a single
// NaN or Inf sample makes every value of that transform - hence the whole returned chunk - NaN, in the call that takes it and in the
// two calls after it (tests/golden/kat_nonfinite.npz, captured from the reference).  The overlap-save kernels would poison the blocks
// whose window holds the sample instead: a subset of those three chunks.  This kernel restores the reference's behaviour for callers
// that want it (the drop-in classes' apply): one workgroup per channel scans the new chunk, records "non-finite" in the channel's
// three-slot flag ring, and overwrites the output chunk with NaN when any of the three slots is set.  One launch, behind the filter
// kernel on the same stream; no host round trip.
#include <hip/hip_runtime.h>

#include "../../include/adsp.h"
#include "capi_common.hpp"

using adsp::fail;

namespace {
constexpr int GUARD_THREADS = 256;

__device__ __forceinline__ unsigned nonfinite_bits(float v) { return (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u ? 1u : 0u; }

__global__ __launch_bounds__(GUARD_THREADS) void nonfinite_guard_kernel(const float* __restrict__ x, float* __restrict__ y, int n,
                                                                         unsigned* __restrict__ flags, int slot) {
    const int c = blockIdx.x;
    const float* xc = x + static_cast<size_t>(c) * n;
    float* yc = y + static_cast<size_t>(c) * n;
    unsigned bad = 0;
    const int n4 = (reinterpret_cast<size_t>(xc) & 15) == 0 ? n / 4 : 0;  // 16-byte loads where the chunk is aligned
    for (int i = threadIdx.x; i < n4; i += GUARD_THREADS) {
        const float4 v = reinterpret_cast<const float4*>(xc)[i];
        bad |= nonfinite_bits(v.x) | nonfinite_bits(v.y) | nonfinite_bits(v.z) | nonfinite_bits(v.w);
    }
    for (int i = 4 * n4 + threadIdx.x; i < n; i += GUARD_THREADS) bad |= nonfinite_bits(xc[i]);
    const int any = __syncthreads_or(static_cast<int>(bad));
    unsigned* f = flags + 3 * c;
    if (threadIdx.x == 0) f[slot] = any ? 1u : 0u;
    const int s1 = slot == 0 ? 2 : slot - 1, s2 = slot == 2 ? 0 : slot + 1;
    const unsigned poisoned = (any ? 1u : 0u) | f[s1] | f[s2];  // the other two slots were written by earlier launches of this stream
    if (!poisoned) return;
    const float qnan = __uint_as_float(0x7fc00000u);
    for (int i = threadIdx.x; i < n; i += GUARD_THREADS) yc[i] = qnan;
}
}  // namespace

extern "C" int adsp_nonfinite_guard(int device_id, const float* d_in, float* d_out, int n_channels, int chunk_size, unsigned* d_flags,
                                    int slot, void* stream) {
    if (!d_in || !d_out || !d_flags) return fail(ADSP_ERR_ARG, "NULL argument");
    if (n_channels < 1 || chunk_size < 1) return fail(ADSP_ERR_ARG, "n_channels and chunk_size must be positive");
    if (slot < 0 || slot > 2) return fail(ADSP_ERR_ARG, "slot must be 0, 1 or 2 (call index modulo 3)");
    HIP_TRY(hipSetDevice(device_id));
    hipLaunchKernelGGL(nonfinite_guard_kernel, dim3((unsigned)n_channels), dim3(GUARD_THREADS), 0, (hipStream_t)stream, d_in, d_out,
                       chunk_size, d_flags, slot);
    HIP_TRY(hipGetLastError());
    return ADSP_OK;
}
