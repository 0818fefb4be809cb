// adsp_exact.hip - exact mode: the streaming FIR as a float64 DIRECT sum on the GPU.
//
//     out[tau] = sum_t taps[t] * s[tau - delay - t]          (per channel, zero history before the first sample)
//
// Why it exists: the reference's WAV front end ends in (y * 32767).astype(int16) - truncation - so two
// float32 FFT pipelines that differ by 1e-7 of full scale disagree by one LSB wherever y * 32767 lands that close to an
// integer (0.01 .. 0.1 % of the samples; the reference itself computes in complex128).  Bit-exactness at the int16 level
// needs the filter value to ~1e-9 of full scale, which no float32 transform gives and a float64 direct sum gives with
// seven orders of magnitude to spare.  WAV files are small (Example1: 264 600 samples x 2047 taps = 1 GFLOP), so the
// O(taps) cost is irrelevant there; MI355X has 78 TFLOP/s of float64 vector math.  The same kernel on float32 batches is
// the on-device ground truth the full-size parity tests check EVERY channel against.
//
// One workgroup = one channel x 1024 consecutive outputs.  Taps are walked in segments of 1024: the segment and the
// 2047 input samples it meets are staged in LDS as doubles, every thread accumulates four outputs (lane-consecutive
// LDS reads: conflict-free, one broadcast read per tap).  Conversions follow the reference to the letter:
//   int16 in : x = float32(pcm) / 32768                      (Utility.py:236-237; exact)
//   int16 out: int16(trunc(float32(y) * float32(32767)))     (astype('float32') at EffectFFTFilter.py:75, Utility.py:306)
//   float32  : out = float32(y)
#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/adsp.h"
#include "capi_common.hpp"

using adsp::fail;

namespace {

constexpr int TILE = 1024, TS = 1024, NT = 256, KPT = TILE / NT;

struct ExactArgs {
    const void* in;      // [n_steps][C][N] samples (float32 or int16)
    void* out;           // [n_steps][C][N]
    const void* hist;    // [C][H] the H most recent past samples of every channel, oldest first
    void* hist_next;     // [C][H] the same after this launch (history update kernel only)
    const double* taps;  // [n_taps]
    int C, N, n_steps, n_taps, delay, H;
};

template <bool S16>
struct Fmt;
template <>
struct Fmt<false> {
    using T = float;
    static __device__ __forceinline__ double to_double(float v) { return static_cast<double>(v); }
    static __device__ __forceinline__ float from_double(double y) { return static_cast<float>(y); }
};
template <>
struct Fmt<true> {
    using T = short;
    static __device__ __forceinline__ double to_double(short v) { return static_cast<double>(static_cast<float>(v) / 32768.0f); }
    static __device__ __forceinline__ short from_double(double y) {
        const float v = static_cast<float>(y) * 32767.0f;  // float32 product, like numpy's float32 array * 32767
        return static_cast<short>(static_cast<int>(v));    // truncation toward zero, low 16 bits
    }
};

// sample of channel c at time tt relative to the start of this launch (tt < 0: history; outside what exists: 0)
template <bool S16>
__device__ __forceinline__ typename Fmt<S16>::T fetch_raw(const ExactArgs& a, int c, long long tt) {
    using T = typename Fmt<S16>::T;
    if (tt < 0) {
        if (tt < -static_cast<long long>(a.H)) return T(0);
        return static_cast<const T*>(a.hist)[static_cast<size_t>(c) * a.H + static_cast<size_t>(a.H + tt)];
    }
    const long long q = tt / a.N;
    if (q >= a.n_steps) return T(0);
    const int r = static_cast<int>(tt - q * a.N);
    return static_cast<const T*>(a.in)[(static_cast<size_t>(q) * a.C + c) * a.N + r];
}

template <bool S16>
__global__ __launch_bounds__(NT) void exact_fir_kernel(const ExactArgs a) {
    using F = Fmt<S16>;
    __shared__ double lt[TS];
    __shared__ double lx[TILE + TS];
    const int tid = static_cast<int>(threadIdx.x);
    const int c = static_cast<int>(blockIdx.y);
    const long long o = static_cast<long long>(blockIdx.x) * TILE;  // first output time of this tile
    double acc[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) acc[k] = 0.0;
    for (int t0 = 0; t0 < a.n_taps; t0 += TS) {
        const int tlen = a.n_taps - t0 < TS ? a.n_taps - t0 : TS;
        __syncthreads();  // the previous segment has been consumed
        for (int i = tid; i < TS; i += NT) lt[i] = i < tlen ? a.taps[t0 + i] : 0.0;
        // lx[j] = s[o - delay - t0 - (TS - 1) + j]: tap t0 + t of output o + i meets lx[(TS - 1) - t + i]
        const long long base = o - a.delay - t0 - (TS - 1);
        for (int j = tid; j < TILE + TS; j += NT) lx[j] = F::to_double(fetch_raw<S16>(a, c, base + j));
        __syncthreads();
        for (int t = 0; t < tlen; ++t) {
            const double w = lt[t];  // same address in every lane: one broadcast read
#pragma unroll
            for (int k = 0; k < KPT; ++k) acc[k] = fma(w, lx[(TS - 1) - t + tid + NT * k], acc[k]);
        }
    }
    const long long total = static_cast<long long>(a.n_steps) * a.N;
    using T = typename F::T;
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const long long tau = o + tid + NT * k;
        if (tau < total) {
            const long long q = tau / a.N;
            const int r = static_cast<int>(tau - q * a.N);
            static_cast<T*>(a.out)[(static_cast<size_t>(q) * a.C + c) * a.N + r] = F::from_double(acc[k]);
        }
    }
}

// hist_next[c][j] = sample at time total - H + j (old history or this launch's input)
template <bool S16>
__global__ __launch_bounds__(NT) void exact_history_kernel(const ExactArgs a) {
    using T = typename Fmt<S16>::T;
    const int c = static_cast<int>(blockIdx.y);
    const int j = static_cast<int>(blockIdx.x) * NT + static_cast<int>(threadIdx.x);
    if (j >= a.H) return;
    const long long total = static_cast<long long>(a.n_steps) * a.N;
    static_cast<T*>(a.hist_next)[static_cast<size_t>(c) * a.H + j] = fetch_raw<S16>(a, c, total - a.H + j);
}

}  // namespace

struct adsp_exact {
    adsp_exact_config cfg;
    int H;            // history samples per channel: delay + n_taps - 1, rounded up to whole chunks
    double* d_taps;
    char* hist[2];    // ping-pong: the update kernel reads one and writes the other
    int cur;
    char* stage_in;
    char* stage_out;
    size_t stage_bytes;
    size_t ssize() const { return cfg.sample_format == ADSP_FORMAT_S16 ? sizeof(short) : sizeof(float); }
};

extern "C" {

int adsp_exact_create(const adsp_exact_config* cfg, const double* taps, adsp_exact** out) {
    if (!cfg || !taps || !out) return fail(ADSP_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (cfg->chunk_size < 1) return fail(ADSP_ERR_ARG, "chunk_size must be positive");
    if (cfg->n_channels < 1 || cfg->n_channels > 65535) return fail(ADSP_ERR_ARG, "n_channels must be in 1..65535");
    if (cfg->n_taps < 1) return fail(ADSP_ERR_ARG, "n_taps must be positive");
    if (cfg->delay < 0) return fail(ADSP_ERR_ARG, "delay %d: a streaming filter cannot look ahead of its input", cfg->delay);
    if (cfg->sample_format != ADSP_FORMAT_F32 && cfg->sample_format != ADSP_FORMAT_S16)
        return fail(ADSP_ERR_ARG, "sample_format %d: need ADSP_FORMAT_F32 or ADSP_FORMAT_S16", cfg->sample_format);
    int ndev = 0;
    int rc = adsp_device_count(&ndev);
    if (rc) return rc;
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(ADSP_ERR_ARG, "device_id %d out of range (%d devices)", cfg->device_id, ndev);
    HIP_TRY(hipSetDevice(cfg->device_id));
    adsp_exact* e = new adsp_exact();
    e->cfg = *cfg;
    const long long reach = (long long)cfg->delay + cfg->n_taps - 1;
    e->H = (int)((reach + cfg->chunk_size - 1) / cfg->chunk_size) * cfg->chunk_size;
    if (e->H < cfg->chunk_size) e->H = cfg->chunk_size;
    e->d_taps = nullptr;
    e->hist[0] = e->hist[1] = nullptr;
    e->cur = 0;
    e->stage_in = e->stage_out = nullptr;
    e->stage_bytes = 0;
    auto bail = [&](int code) {
        adsp_exact_destroy(e);
        return code;
    };
    hipError_t err;
    if ((err = hipMalloc(&e->d_taps, (size_t)cfg->n_taps * sizeof(double))) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMalloc: %s", hipGetErrorString(err)));
    if ((err = hipMemcpy(e->d_taps, taps, (size_t)cfg->n_taps * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(err)));
    const size_t hb = (size_t)cfg->n_channels * e->H * e->ssize();
    for (int i = 0; i < 2; ++i) {
        if ((err = hipMalloc(&e->hist[i], hb)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMalloc history (%zu bytes): %s", hb, hipGetErrorString(err)));
        if ((err = hipMemset(e->hist[i], 0, hb)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMemset: %s", hipGetErrorString(err)));
    }
    *out = e;
    return ADSP_OK;
}

void adsp_exact_destroy(adsp_exact* e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device_id);
    (void)hipDeviceSynchronize();
    for (void* p : {(void*)e->d_taps, (void*)e->hist[0], (void*)e->hist[1], (void*)e->stage_in, (void*)e->stage_out})
        if (p) (void)hipFree(p);
    delete e;
}

int adsp_exact_reset(adsp_exact* e) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    HIP_TRY(hipSetDevice(e->cfg.device_id));
    HIP_TRY(hipDeviceSynchronize());
    for (int i = 0; i < 2; ++i) HIP_TRY(hipMemset(e->hist[i], 0, (size_t)e->cfg.n_channels * e->H * e->ssize()));
    return ADSP_OK;
}

int adsp_exact_apply_device(adsp_exact* e, const void* d_in, void* d_out, int n_steps, void* stream_v) {
    if (!e || !d_in || !d_out) return fail(ADSP_ERR_ARG, "NULL argument");
    if (n_steps <= 0) return fail(ADSP_ERR_ARG, "n_steps must be positive");
    if (d_in == d_out) return fail(ADSP_ERR_ARG, "in-place is not supported: outputs of one tile are inputs of the next");
    HIP_TRY(hipSetDevice(e->cfg.device_id));
    hipStream_t stream = (hipStream_t)stream_v;
    ExactArgs a;
    a.in = d_in;
    a.out = d_out;
    a.hist = e->hist[e->cur];
    a.hist_next = e->hist[e->cur ^ 1];
    a.taps = e->d_taps;
    a.C = e->cfg.n_channels;
    a.N = e->cfg.chunk_size;
    a.n_steps = n_steps;
    a.n_taps = e->cfg.n_taps;
    a.delay = e->cfg.delay;
    a.H = e->H;
    const long long total = (long long)n_steps * a.N;
    const long long tiles = (total + TILE - 1) / TILE;
    if (tiles > 0x7fffffffLL) return fail(ADSP_ERR_ARG, "n_steps %d x chunk %d is too long for one call; split it", n_steps, a.N);
    const dim3 grid((unsigned)tiles, (unsigned)a.C), hgrid((unsigned)((a.H + NT - 1) / NT), (unsigned)a.C);
    if (e->cfg.sample_format == ADSP_FORMAT_S16) {
        hipLaunchKernelGGL(exact_fir_kernel<true>, grid, dim3(NT), 0, stream, a);
        hipLaunchKernelGGL(exact_history_kernel<true>, hgrid, dim3(NT), 0, stream, a);
    } else {
        hipLaunchKernelGGL(exact_fir_kernel<false>, grid, dim3(NT), 0, stream, a);
        hipLaunchKernelGGL(exact_history_kernel<false>, hgrid, dim3(NT), 0, stream, a);
    }
    HIP_TRY(hipGetLastError());
    e->cur ^= 1;
    return ADSP_OK;
}

int adsp_exact_apply_host(adsp_exact* e, const void* in, void* out, int n_steps) {
    if (!e || !in || !out) return fail(ADSP_ERR_ARG, "NULL argument");
    if (n_steps <= 0) return fail(ADSP_ERR_ARG, "n_steps must be positive");
    HIP_TRY(hipSetDevice(e->cfg.device_id));
    const size_t bytes = (size_t)n_steps * e->cfg.n_channels * e->cfg.chunk_size * e->ssize();
    if (bytes > e->stage_bytes) {
        HIP_TRY(hipDeviceSynchronize());
        if (e->stage_in) (void)hipFree(e->stage_in);
        if (e->stage_out) (void)hipFree(e->stage_out);
        e->stage_in = e->stage_out = nullptr;
        e->stage_bytes = 0;
        HIP_TRY(hipMalloc(&e->stage_in, bytes));
        HIP_TRY(hipMalloc(&e->stage_out, bytes));
        e->stage_bytes = bytes;
    }
    HIP_TRY(hipMemcpy(e->stage_in, in, bytes, hipMemcpyHostToDevice));
    int rc = adsp_exact_apply_device(e, e->stage_in, e->stage_out, n_steps, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, e->stage_out, bytes, hipMemcpyDeviceToHost));
    return ADSP_OK;
}

}  // extern "C"
