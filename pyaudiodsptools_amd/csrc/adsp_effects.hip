// adsp_effects.hip - the standalone elementwise passes behind include/adsp.h: the fused output effects as kernels of their own
// (adsp_effect_*), the tremolo over rows, MixSignals (adsp_mix_*), the reference's treatment of non-finite samples
// (adsp_nonfinite_guard) and the shader-clock probe of bench.py.  No engine object here.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "../../include/adsp.h"
#include "capi_common.hpp"
#include "plan_table.hpp"

using adsp::fail;

namespace adsp {
namespace {
constexpr int kMaxWindows = 64;
HostWindow g_windows[kMaxWindows];  // never freed: pinned memory of a process that is about to exit is the driver's to release
}  // namespace

HostWindow* host_window(int device_id) {
    int ndev = 0;
    if (adsp_device_count(&ndev)) return nullptr;
    if (device_id < 0 || device_id >= ndev || device_id >= kMaxWindows) {
        fail(ADSP_ERR_ARG, "device_id %d out of range (%d devices)", device_id, ndev);
        return nullptr;
    }
    return &g_windows[device_id];
}

int host_window_reserve(HostWindow& w, size_t in_bytes, size_t out_bytes) {
    struct Side { char** host; void** dev; size_t* cap; size_t want; } sides[2] = {{&w.in, &w.d_in, &w.cap_in, in_bytes}, {&w.out, &w.d_out, &w.cap_out, out_bytes}};
    for (Side& s : sides) {
        if (s.want <= *s.cap) continue;
        if (*s.host) {
            HIP_TRY(hipDeviceSynchronize());  // (calls hold the mutex until they are done: nothing reads the old buffer, this is belt and braces)
            (void)hipHostFree(*s.host);
            *s.host = nullptr;
            *s.cap = 0;
        }
        size_t cap = size_t(64) << 10;
        while (cap < s.want) cap *= 2;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(s.host), cap, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer(s.dev, *s.host, 0));
        *s.cap = cap;
    }
    if (!w.done) HIP_TRY(hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
    return ADSP_OK;
}

int host_window_wait(HostWindow& w, hipStream_t stream) {
    HIP_TRY(hipEventRecord(w.done, stream));
    HIP_TRY(hipEventSynchronize(w.done));
    return ADSP_OK;
}
}  // namespace adsp

// standalone elementwise form of the fused output effects (fftconv_kernel.hpp::epilogue_value): out[i] = effect(in[i]);
// the tremolo multiplies by its periodic LFO table, element 0 at table index `phase`
__global__ void adsp_pointwise_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n, int op, float p0,
                                      float p1, float p2, int phase) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const size_t i0 = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (op == ADSP_EFFECT_TREMOLO) {
        // table index of this thread's first element, then advanced by (stride mod len) per iteration: one 64-bit
        // modulo per thread instead of one per sample
        const unsigned len = static_cast<unsigned>(p2);
        unsigned idx = static_cast<unsigned>((static_cast<unsigned long long>(phase) + i0) % len);
        const unsigned step = static_cast<unsigned>(stride % len);
        size_t i = i0;
        for (; i + 3 * stride < n; i += 4 * stride) {  // four loads in flight per lane
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = in[i + u * stride];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                out[i + u * stride] = v[u] * adsp::tremolo_gain(static_cast<int>(idx), p0, p1);
                idx += step;
                idx -= idx >= len ? len : 0;
            }
        }
        for (; i < n; i += stride) {
            out[i] = in[i] * adsp::tremolo_gain(static_cast<int>(idx), p0, p1);
            idx += step;
            idx -= idx >= len ? len : 0;
        }
        return;
    }
    size_t i = i0;
    for (; i + 3 * stride < n; i += 4 * stride) {  // four loads in flight per lane (5.1 -> 5.7 TB/s measured)
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) out[i + u * stride] = adsp::epilogue_value(v[u], op, p0, p1, p2);
    }
    for (; i < n; i += stride) out[i] = adsp::epilogue_value(in[i], op, p0, p1, p2);
}

// The tremolo over a [rows][row_len] batch whose every row is one channel's chunk: all rows start at LFO table index `phase` (the
// reference runs one tremolo device per channel, all in step: EffectTremolo.py:40-46).  blockIdx.y = row.
__global__ void adsp_tremolo_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int row_len, float depth, float rev_per_sample,
                                         int len, int phase) {
    const size_t row = static_cast<size_t>(blockIdx.y) * row_len;
    const float inv_len = 1.f / static_cast<float>(len);
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < row_len; r += gridDim.x * blockDim.x) {
        int idx = phase + r;  // < 2^24 + 2^23
        idx -= static_cast<int>(static_cast<float>(idx) * inv_len) * len;
        idx += idx < 0 ? len : 0;
        idx -= idx >= len ? len : 0;
        out[row + r] = in[row + r] * adsp::tremolo_gain(idx, depth, rev_per_sample);
    }
}

// MixSignals (Utility.py:51-72): out = clip(sum of k signals) - up to 8 addends per pass
struct MixArgs {
    const float* in[8];
    int k;
    int add_existing;  // out already holds a partial sum
    int clip;
};
__global__ void adsp_mix_kernel(MixArgs a, float* __restrict__ out, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    for (; i + stride < n; i += 2 * stride) {  // two elements per iteration: twice the loads in flight per lane
        float acc0 = a.add_existing ? out[i] : 0.f, acc1 = a.add_existing ? out[i + stride] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < a.k) {
                acc0 += a.in[j][i];
                acc1 += a.in[j][i + stride];
            }
        out[i] = a.clip ? __builtin_amdgcn_fmed3f(acc0, -1.f, 1.f) : acc0;
        out[i + stride] = a.clip ? __builtin_amdgcn_fmed3f(acc1, -1.f, 1.f) : acc1;
    }
    for (; i < n; i += stride) {
        float acc = a.add_existing ? out[i] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < a.k) acc += a.in[j][i];
        out[i] = a.clip ? __builtin_amdgcn_fmed3f(acc, -1.f, 1.f) : acc;
    }
}

// Shader clock while a workload runs (bench.py): one lane counts shader cycles (s_memtime) over a stretch of the constant
// 100 MHz clock (s_memrealtime), sleeping between reads - launched on a side stream next to the timed kernels.
__global__ void adsp_clock_probe_kernel(unsigned long long* out, unsigned long long ticks) {
    const unsigned long long w0 = wall_clock64();
    const unsigned long long c0 = clock64();
    unsigned long long w1 = w0;
    while (w1 - w0 < ticks) {
        __builtin_amdgcn_s_sleep(64);
        w1 = wall_clock64();
    }
    out[0] = clock64() - c0;
    out[1] = w1 - w0;
}

extern "C" {

namespace {
int pointwise_launch(int device_id, int effect, float p0, float p1, float p2, int phase, const float* d_in, float* d_out,
                     size_t n, hipStream_t stream) {
    if (effect < ADSP_EFFECT_NONE || effect > ADSP_EFFECT_BIT_CRUSHER) return fail(ADSP_ERR_ARG, "unknown effect %d", effect);
    if (effect == ADSP_EFFECT_TREMOLO && (!(p2 >= 1.f && p2 <= 8388608.f) || phase < 0 || phase >= (int)p2))
        return fail(ADSP_ERR_ARG, "tremolo: p2 = table length (1..2^23), 0 <= phase < p2");
    HIP_TRY(hipSetDevice(device_id));
    if (n == 0) return ADSP_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;  // grid-stride beyond 8 workgroups per CU
    hipLaunchKernelGGL(adsp_pointwise_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_in, d_out, n, effect, p0, p1, p2, phase);
    HIP_TRY(hipGetLastError());
    return ADSP_OK;
}
}  // namespace

int adsp_effect_device(int device_id, int effect, float p0, float p1, float p2, int phase, const float* d_in,
                       float* d_out, size_t n, void* stream) {
    if (!d_in || !d_out) return fail(ADSP_ERR_ARG, "NULL argument");
    return pointwise_launch(device_id, effect, p0, p1, p2, phase, d_in, d_out, n, (hipStream_t)stream);
}

int adsp_tremolo_rows_device(int device_id, float depth, float lfo_per_sample, int lfo_length, int phase, const float* d_in, float* d_out, int rows,
                             int row_len, void* stream) {
    if (!d_in || !d_out) return fail(ADSP_ERR_ARG, "NULL argument");
    if (rows < 1 || rows > 65535 || row_len < 1 || row_len > (1 << 24)) return fail(ADSP_ERR_ARG, "rows 1..65535 of 1..2^24 samples");
    if (lfo_length < 1 || lfo_length > (1 << 23) || phase < 0 || phase >= lfo_length) return fail(ADSP_ERR_ARG, "tremolo: table length 1..2^23, 0 <= phase < length");
    HIP_TRY(hipSetDevice(device_id));
    unsigned bx = (unsigned)((row_len + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(adsp_tremolo_rows_kernel, dim3(bx, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, d_in, d_out, row_len, depth, lfo_per_sample,
                       lfo_length, phase);
    HIP_TRY(hipGetLastError());
    return ADSP_OK;
}

int adsp_mix_device(int device_id, const float* const* d_inputs, int k, int clip, float* d_out, size_t n, void* stream) {
    if (!d_inputs || !d_out) return fail(ADSP_ERR_ARG, "NULL argument");
    if (k < 1) return fail(ADSP_ERR_ARG, "mix needs at least one input");
    for (int j = 0; j < k; ++j)
        if (!d_inputs[j]) return fail(ADSP_ERR_ARG, "NULL input %d", j);
    HIP_TRY(hipSetDevice(device_id));
    if (n == 0) return ADSP_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    for (int j0 = 0; j0 < k; j0 += 8) {
        MixArgs a;
        a.k = k - j0 < 8 ? k - j0 : 8;
        for (int j = 0; j < 8; ++j) a.in[j] = j < a.k ? d_inputs[j0 + j] : nullptr;
        a.add_existing = j0 > 0;
        a.clip = (clip && j0 + 8 >= k) ? 1 : 0;
        hipLaunchKernelGGL(adsp_mix_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, d_out, n);
        HIP_TRY(hipGetLastError());
    }
    return ADSP_OK;
}

int adsp_mix_host(int device_id, const float* const* inputs, int k, int clip, float* out, size_t n) {
    if (!inputs || !out) return fail(ADSP_ERR_ARG, "NULL argument");
    if (k < 1) return fail(ADSP_ERR_ARG, "mix needs at least one input");
    for (int j = 0; j < k; ++j)
        if (!inputs[j]) return fail(ADSP_ERR_ARG, "NULL input %d", j);
    int ndev = 0;
    int rc = adsp_device_count(&ndev);
    if (rc) return rc;
    if (device_id < 0 || device_id >= ndev) return fail(ADSP_ERR_ARG, "device_id %d out of range (%d devices)", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    if (n == 0) return ADSP_OK;
    if ((size_t)k * n * sizeof(float) <= adsp::kHostWindowMax) {  // small: the kernel reads the inputs from pinned host memory
        adsp::HostWindow* w = adsp::host_window(device_id);
        if (!w) return ADSP_ERR_ARG;
        std::lock_guard<std::mutex> lock(w->mu);
        if ((rc = adsp::host_window_reserve(*w, (size_t)k * n * sizeof(float), n * sizeof(float)))) return rc;
        std::vector<const float*> win(k);
        for (int j = 0; j < k; ++j) {
            memcpy(w->in + (size_t)j * n * sizeof(float), inputs[j], n * sizeof(float));
            win[j] = static_cast<const float*>(w->d_in) + (size_t)j * n;
        }
        if ((rc = adsp_mix_device(device_id, win.data(), k, clip, static_cast<float*>(w->d_out), n, nullptr))) return rc;
        if ((rc = adsp::host_window_wait(*w, nullptr))) return rc;
        memcpy(out, w->out, n * sizeof(float));
        return ADSP_OK;
    }
    float* d = nullptr;  // [k + 1][n]: the inputs, then the sum
    HIP_TRY(hipMalloc(&d, (size_t)(k + 1) * n * sizeof(float)));
    std::vector<const float*> ptrs(k);
    hipError_t err = hipSuccess;
    for (int j = 0; j < k && err == hipSuccess; ++j) {
        ptrs[j] = d + (size_t)j * n;
        err = hipMemcpy(d + (size_t)j * n, inputs[j], n * sizeof(float), hipMemcpyHostToDevice);
    }
    if (err == hipSuccess) {
        rc = adsp_mix_device(device_id, ptrs.data(), k, clip, d + (size_t)k * n, n, nullptr);
        if (rc == ADSP_OK) err = hipMemcpy(out, d + (size_t)k * n, n * sizeof(float), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d);
    if (rc) return rc;
    if (err != hipSuccess) return fail(ADSP_ERR_HIP, "mix copy failed: %s", hipGetErrorString(err));
    return ADSP_OK;
}

int adsp_effect_host(int device_id, int effect, float p0, float p1, float p2, int phase, const float* in, float* out,
                     size_t n) {
    if (!in || !out) return fail(ADSP_ERR_ARG, "NULL argument");
    int ndev = 0;
    int rc = adsp_device_count(&ndev);
    if (rc) return rc;
    if (device_id < 0 || device_id >= ndev) return fail(ADSP_ERR_ARG, "device_id %d out of range (%d devices)", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    if (n == 0) return ADSP_OK;
    if (n * sizeof(float) <= adsp::kHostWindowMax) {  // a chunk or a few: no allocation, no staging copy (capi_common.hpp)
        adsp::HostWindow* w = adsp::host_window(device_id);
        if (!w) return ADSP_ERR_ARG;
        std::lock_guard<std::mutex> lock(w->mu);
        if ((rc = adsp::host_window_reserve(*w, n * sizeof(float), n * sizeof(float)))) return rc;
        memcpy(w->in, in, n * sizeof(float));
        if ((rc = pointwise_launch(device_id, effect, p0, p1, p2, phase, static_cast<const float*>(w->d_in), static_cast<float*>(w->d_out), n, nullptr))) return rc;
        if ((rc = adsp::host_window_wait(*w, nullptr))) return rc;
        memcpy(out, w->out, n * sizeof(float));
        return ADSP_OK;
    }
    float* d = nullptr;
    HIP_TRY(hipMalloc(&d, n * sizeof(float)));
    hipError_t err = hipMemcpy(d, in, n * sizeof(float), hipMemcpyHostToDevice);
    if (err == hipSuccess) {
        rc = pointwise_launch(device_id, effect, p0, p1, p2, phase, d, d, n, nullptr);
        if (rc == ADSP_OK) err = hipMemcpy(out, d, n * sizeof(float), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d);
    if (rc) return rc;
    if (err != hipSuccess) return fail(ADSP_ERR_HIP, "effect copy failed: %s", hipGetErrorString(err));
    return ADSP_OK;
}

int adsp_clock_probe_launch(int device_id, double microseconds, void* stream, unsigned long long** result) {
    if (!result) return fail(ADSP_ERR_ARG, "result is NULL");
    if (!(microseconds > 0.0) || microseconds > 1e6) return fail(ADSP_ERR_ARG, "probe length must be in (0, 1e6] us");
    HIP_TRY(hipSetDevice(device_id));
    unsigned long long* host = nullptr;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&host), 2 * sizeof(unsigned long long), hipHostMallocMapped));
    host[0] = host[1] = 0;
    void* dptr = nullptr;
    hipError_t err = hipHostGetDevicePointer(&dptr, host, 0);
    if (err != hipSuccess) {
        (void)hipHostFree(host);
        return fail(ADSP_ERR_HIP, "hipHostGetDevicePointer: %s", hipGetErrorString(err));
    }
    hipLaunchKernelGGL(adsp_clock_probe_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, static_cast<unsigned long long*>(dptr),
                       (unsigned long long)(microseconds * 100.0));
    err = hipGetLastError();
    if (err != hipSuccess) {
        (void)hipHostFree(host);
        return fail(ADSP_ERR_HIP, "clock probe launch: %s", hipGetErrorString(err));
    }
    *result = host;
    return ADSP_OK;
}

int adsp_clock_probe_read(int device_id, void* stream, unsigned long long* result, double* shader_mhz) {
    if (!result || !shader_mhz) return fail(ADSP_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(device_id));
    hipError_t err = hipStreamSynchronize((hipStream_t)stream);
    const unsigned long long cycles = result[0], ticks = result[1];
    (void)hipHostFree(result);
    if (err != hipSuccess) return fail(ADSP_ERR_HIP, "hipStreamSynchronize: %s", hipGetErrorString(err));
    if (!ticks) return fail(ADSP_ERR_STATE, "the clock probe has not run");
    *shader_mhz = (double)cycles / (double)ticks * 100.0;
    return ADSP_OK;
}

}  // extern "C"

// ---- non-finite inputs the way the reference treats them (include/adsp.h: adsp_nonfinite_guard) ----
//
// The reference transforms chunks k-2, k-1, k as ONE 3N-point buffer (EffectFFTFilter.py:67-72, EffectEQ3BandFFT.py:175-179): a single
// NaN or Inf sample makes every value of that transform - hence the whole returned chunk - NaN, in the call that takes it and in the
// two calls after it (tests/golden/kat_nonfinite.npz, captured from the reference).  The overlap-save kernels would poison the blocks
// whose window holds the sample instead: a subset of those three chunks.  This kernel restores the reference's behaviour for callers
// that want it (the drop-in classes' apply): one workgroup per channel scans the new chunk, records "non-finite" in the channel's
// three-slot flag ring, and overwrites the output chunk with NaN when any of the three slots is set.  One launch, behind the filter
// kernel on the same stream; no host round trip.
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
