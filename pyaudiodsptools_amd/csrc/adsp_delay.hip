// adsp_delay.hip - tapped delay line engine: the arithmetic of the reference's CreateDelay
// (EffectDelay.py:60-72) and of its reverb's delay lines (_EffectReverb.py:46-58).
//
// Both add the chunk, scaled by gain k, into an accumulation buffer spacing*(k+1) samples ahead and return the head of
// that buffer (plus the chunk itself unless `wet`).  That is a sparse FIR,
//     out[t] = dry * x[t] + sum_k gain[k] * x[t - delay[k]],
// so the GPU form keeps the INPUT history (a ring of past chunks, like the FFT engine) and gathers: one thread = four
// consecutive outputs, one unaligned 16-byte load per tap.  HBM-bound: (taps + 2) * 4 bytes per sample when the taps
// reach further back than the caches hold, less otherwise.
#include <hip/hip_runtime.h>

#include <cstring>

#include <vector>

#include "../../include/adsp.h"
#include "capi_common.hpp"

using adsp::fail;

namespace {

struct DelayArgs {
    const float* ring;  // [slots][C][N] input history
    const float* in;    // [n_steps][C][N]
    float* out;         // [n_steps][C][N]
    const int* delay;   // [K] samples, 1 <= delay <= H * N
    const float* gain;  // [K]
    int K;
    float dry;
    int ring_pos, ring_slots;
    int C, N, H;  // H = history chunks
    float inv_n;
    int quad_blocks;  // workgroups per [channel] row: ceil(N / 4 / 256)
    int accumulate;   // add to what `out` holds (second line of the reverb)
};

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ const float* chunk_base(const DelayArgs& a, int q, size_t chan_off, size_t plane) {
    if (q >= 0) return a.in + static_cast<size_t>(q) * plane + chan_off;
    int slot = a.ring_pos + 1 + q;  // q >= -H, ring_slots >= H
    slot += slot < 0 ? a.ring_slots : 0;
    return a.ring + static_cast<size_t>(slot) * plane + chan_off;
}

__global__ __launch_bounds__(256) void delay_kernel(const DelayArgs a) {
#pragma clang fp contract(off)
    // blockIdx.x = channel * quad_blocks + block of 256 quads; blockIdx.y = step
    const int c = static_cast<int>(blockIdx.x) / a.quad_blocks;  // wave-uniform
    const int qb = static_cast<int>(blockIdx.x) - c * a.quad_blocks;
    const int s = static_cast<int>(blockIdx.y);
    const int i = (qb * 256 + static_cast<int>(threadIdx.x)) * 4;
    if (i >= a.N) return;
    const size_t plane = static_cast<size_t>(a.C) * a.N;
    const size_t chan_off = static_cast<size_t>(c) * a.N;
    const float* cur = a.in + static_cast<size_t>(s) * plane + chan_off + i;
    const v4f x = *reinterpret_cast<const v4f*>(cur);
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    // Same float32 operations in the same order as the reference whenever its additions have one order: every tap is
    // a separate multiply and add (no fma), contributions that entered its buffer first - the longest delays - are
    // summed first, the chunk itself is added last (EffectDelay.py:60-67).  Bit-exact when taps are >= one chunk apart.
    for (int k = a.K - 1; k >= 0; --k) {
        const int d = a.delay[k];  // scalar loads
        const float g = a.gain[k];
        // source time of element 0 on an axis biased by H chunks (>= 0): which chunk, which offset
        const int tb = i - d + a.H * a.N;
        int q = static_cast<int>(static_cast<float>(tb) * a.inv_n);
        int r = tb - q * a.N;
        if (r < 0) {
            --q;
            r += a.N;
        } else if (r >= a.N) {
            ++q;
            r -= a.N;
        }
        q += s - a.H;
        v4f v;
        if (r <= a.N - 4) {  // the usual case: all four samples in one chunk (4-byte aligned 16-byte load)
            __builtin_memcpy(&v, chunk_base(a, q, chan_off, plane) + r, sizeof v);
        } else {  // the quad straddles a chunk boundary
            const float* lo = chunk_base(a, q, chan_off, plane);
            const float* hi = chunk_base(a, q + 1, chan_off, plane);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = r + j < a.N ? lo[r + j] : hi[r + j - a.N];
        }
        acc += g * v;
    }
    v4f y = a.dry * x + acc;
    float* dst = a.out + static_cast<size_t>(s) * plane + chan_off + i;
    if (a.accumulate) y += *reinterpret_cast<const v4f*>(dst);
    __builtin_nontemporal_store(y, reinterpret_cast<v4f*>(dst));
}

}  // namespace

struct adsp_delay {
    adsp_delay_config cfg;
    int H;  // history chunks = ceil(max delay / N), >= 1
    float dry;
    int accumulate;
    float* ring;
    int ring_pos;
    int* d_delay;
    float* d_gain;
    float *stage_in, *stage_out;
    size_t stage_elems;
    size_t plane() const { return (size_t)cfg.n_channels * cfg.chunk_size; }
};

extern "C" {

int adsp_delay_create(const adsp_delay_config* cfg, const int* tap_delay, const float* tap_gain, float dry_gain,
                      adsp_delay** out) {
    if (!cfg || !out) return fail(ADSP_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (cfg->n_taps < 0 || cfg->n_taps > ADSP_DELAY_MAX_TAPS) return fail(ADSP_ERR_ARG, "n_taps %d: need 0..%d", cfg->n_taps, ADSP_DELAY_MAX_TAPS);
    if (cfg->n_taps > 0 && (!tap_delay || !tap_gain)) return fail(ADSP_ERR_ARG, "NULL tap table");
    if (cfg->chunk_size < 4 || cfg->chunk_size % 4) return fail(ADSP_ERR_ARG, "chunk_size %d: need a multiple of 4", cfg->chunk_size);
    if (cfg->n_channels < 1) return fail(ADSP_ERR_ARG, "n_channels must be positive");
    long long max_d = 0;
    for (int k = 0; k < cfg->n_taps; ++k) {
        if (tap_delay[k] < 1) return fail(ADSP_ERR_ARG, "tap %d: delay %d must be at least one sample", k, tap_delay[k]);
        if (tap_delay[k] > max_d) max_d = tap_delay[k];
    }
    const long long N = cfg->chunk_size;
    const long long H = max_d ? (max_d + N - 1) / N : 1;
    if ((H + 1) * N >= (1LL << 24)) return fail(ADSP_ERR_ARG, "longest delay %lld samples: need less than 2^24 - chunk", max_d);
    int ndev = 0;
    int rc = adsp_device_count(&ndev);
    if (rc) return rc;
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(ADSP_ERR_ARG, "device_id %d out of range (%d devices)", cfg->device_id, ndev);
    HIP_TRY(hipSetDevice(cfg->device_id));
    adsp_delay* e = new adsp_delay();
    e->cfg = *cfg;
    e->H = (int)H;
    e->dry = dry_gain;
    e->accumulate = 0;
    e->ring = nullptr;
    e->d_delay = nullptr;
    e->d_gain = nullptr;
    e->stage_in = e->stage_out = nullptr;
    e->stage_elems = 0;
    e->ring_pos = e->H - 1;
    auto bail = [&](int code) {
        adsp_delay_destroy(e);
        return code;
    };
    const size_t ring_bytes = (size_t)e->H * e->plane() * sizeof(float);
    hipError_t err;
    if ((err = hipMalloc(&e->ring, ring_bytes)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMalloc delay ring (%zu bytes): %s", ring_bytes, hipGetErrorString(err)));
    if ((err = hipMemset(e->ring, 0, ring_bytes)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMemset: %s", hipGetErrorString(err)));
    const size_t kb = (size_t)(cfg->n_taps ? cfg->n_taps : 1);
    if ((err = hipMalloc(&e->d_delay, kb * sizeof(int))) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMalloc: %s", hipGetErrorString(err)));
    if ((err = hipMalloc(&e->d_gain, kb * sizeof(float))) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMalloc: %s", hipGetErrorString(err)));
    if (cfg->n_taps) {
        if ((err = hipMemcpy(e->d_delay, tap_delay, cfg->n_taps * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(err)));
        if ((err = hipMemcpy(e->d_gain, tap_gain, cfg->n_taps * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(err)));
    }
    *out = e;
    return ADSP_OK;
}

void adsp_delay_destroy(adsp_delay* e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device_id);
    (void)hipDeviceSynchronize();
    if (e->ring) (void)hipFree(e->ring);
    if (e->d_delay) (void)hipFree(e->d_delay);
    if (e->d_gain) (void)hipFree(e->d_gain);
    if (e->stage_in) (void)hipFree(e->stage_in);
    if (e->stage_out) (void)hipFree(e->stage_out);
    delete e;
}

int adsp_delay_reset(adsp_delay* e) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    HIP_TRY(hipSetDevice(e->cfg.device_id));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(e->ring, 0, (size_t)e->H * e->plane() * sizeof(float)));
    e->ring_pos = e->H - 1;
    return ADSP_OK;
}

int adsp_delay_set_accumulate(adsp_delay* e, int on) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    e->accumulate = on != 0;
    return ADSP_OK;
}

int adsp_delay_history_chunks(const adsp_delay* e, int* chunks) {
    if (!e || !chunks) return fail(ADSP_ERR_ARG, "NULL argument");
    *chunks = e->H;
    return ADSP_OK;
}

int adsp_delay_apply_device(adsp_delay* e, const float* d_in, float* d_out, int n_steps, void* stream_v) {
    if (!e || !d_in || !d_out) return fail(ADSP_ERR_ARG, "NULL argument");
    if (n_steps <= 0) return fail(ADSP_ERR_ARG, "n_steps must be positive");
    if (n_steps > 65535) return fail(ADSP_ERR_ARG, "n_steps %d: at most 65535 per call", n_steps);
    const size_t plane = e->plane();
    const float* in_end = d_in + (size_t)n_steps * plane;
    const float* out_end = d_out + (size_t)n_steps * plane;
    if (d_in < out_end && d_out < in_end) return fail(ADSP_ERR_ARG, "the delay line cannot run in place: taps read earlier input");
    HIP_TRY(hipSetDevice(e->cfg.device_id));
    hipStream_t stream = (hipStream_t)stream_v;
    DelayArgs a;
    a.ring = e->ring;
    a.in = d_in;
    a.out = d_out;
    a.delay = e->d_delay;
    a.gain = e->d_gain;
    a.K = e->cfg.n_taps;
    a.dry = e->dry;
    a.ring_pos = e->ring_pos;
    a.ring_slots = e->H;
    a.C = e->cfg.n_channels;
    a.N = e->cfg.chunk_size;
    a.H = e->H;
    a.inv_n = 1.0f / (float)e->cfg.chunk_size;
    a.quad_blocks = (e->cfg.chunk_size / 4 + 255) / 256;
    a.accumulate = e->accumulate;
    const long long gx = (long long)a.C * a.quad_blocks;
    if (gx > 0x7fffffffLL) return fail(ADSP_ERR_ARG, "launch too large (%lld workgroups per step)", gx);
    hipLaunchKernelGGL(delay_kernel, dim3((unsigned)gx, (unsigned)n_steps), dim3(256), 0, stream, a);
    HIP_TRY(hipGetLastError());
    // the newest min(n_steps, H) chunks replace the oldest ring slots (stream-ordered after the kernel that reads them)
    const int cnt = n_steps < e->H ? n_steps : e->H;
    for (int i = 0; i < cnt; ++i) {
        const int slot = (e->ring_pos + 1 + i) % e->H;
        HIP_TRY(hipMemcpyAsync(e->ring + (size_t)slot * plane, d_in + (size_t)(n_steps - cnt + i) * plane, plane * sizeof(float),
                               hipMemcpyDeviceToDevice, stream));
    }
    e->ring_pos = (e->ring_pos + cnt) % e->H;
    return ADSP_OK;
}

int adsp_delay_apply_host(adsp_delay* e, const float* in, float* out, int n_steps) {
    if (!e || !in || !out) return fail(ADSP_ERR_ARG, "NULL argument");
    if (n_steps <= 0) return fail(ADSP_ERR_ARG, "n_steps must be positive");
    HIP_TRY(hipSetDevice(e->cfg.device_id));
    const size_t elems = (size_t)n_steps * e->plane();
    if (elems * sizeof(float) <= adsp::kHostWindowMax) {  // a chunk or a few: the kernel reads / writes pinned host memory (capi_common.hpp)
        adsp::HostWindow* w = adsp::host_window(e->cfg.device_id);
        if (!w) return ADSP_ERR_ARG;
        std::lock_guard<std::mutex> lock(w->mu);
        int rc = adsp::host_window_reserve(*w, elems * sizeof(float), elems * sizeof(float));
        if (rc) return rc;
        memcpy(w->in, in, elems * sizeof(float));
        if (e->accumulate) memcpy(w->out, out, elems * sizeof(float));
        if ((rc = adsp_delay_apply_device(e, static_cast<const float*>(w->d_in), static_cast<float*>(w->d_out), n_steps, nullptr))) return rc;
        if ((rc = adsp::host_window_wait(*w, nullptr))) return rc;
        memcpy(out, w->out, elems * sizeof(float));
        return ADSP_OK;
    }
    if (elems > e->stage_elems) {
        HIP_TRY(hipDeviceSynchronize());
        if (e->stage_in) (void)hipFree(e->stage_in);
        if (e->stage_out) (void)hipFree(e->stage_out);
        e->stage_in = e->stage_out = nullptr;
        e->stage_elems = 0;
        HIP_TRY(hipMalloc(&e->stage_in, elems * sizeof(float)));
        HIP_TRY(hipMalloc(&e->stage_out, elems * sizeof(float)));
        e->stage_elems = elems;
    }
    HIP_TRY(hipMemcpy(e->stage_in, in, elems * sizeof(float), hipMemcpyHostToDevice));
    if (e->accumulate) HIP_TRY(hipMemcpy(e->stage_out, out, elems * sizeof(float), hipMemcpyHostToDevice));
    int rc = adsp_delay_apply_device(e, e->stage_in, e->stage_out, n_steps, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, e->stage_out, elems * sizeof(float), hipMemcpyDeviceToHost));
    return ADSP_OK;
}

}  // extern "C"
