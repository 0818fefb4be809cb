// adsp_capi.hip - C ABI (include/adsp.h) over the fused overlap-save kernel.
// Host logic only: plan registry, twiddle / spectrum-pair tables (float64 -> float32), the input
// history ring, launches.  There is deliberately NO CPU fallback: without a GPU every compute
// entry point fails with ADSP_ERR_NO_DEVICE / ADSP_ERR_HIP.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/adsp.h"
#include "capi_common.hpp"
#include "plan_table.hpp"
#include "table_build.hpp"

namespace adsp {  // adsp_rccl.hip
int rccl_broadcast(float* const* d_buf, const int* devs, const hipStream_t* streams, int n, size_t count, int root);
int rccl_version(int* version);
int rccl_unique_id(char* out);
int rccl_broadcast_rank(const char* unique_id, int rank, int world, int root, int dev, float* d_buf, size_t count, hipStream_t stream);
int rccl_finalize();
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

namespace {

thread_local std::string g_last_error;
}  // namespace

// shared with the other translation units of the library (capi_common.hpp)
int adsp::fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

namespace {
using adsp::fail;

using adsp::PlanInfo;
using namespace adsp::tables;

const PlanInfo* find_plan(int M, int FQ, int fmt) {
    int n = 0;
#ifdef ADSP_TUNING_BUILD  // the A/B plans of plans_var.hip are linked into libadsp_tuning.so only (make tuning)
    if (fmt == ADSP_FORMAT_F32) {
        const char* v = getenv("ADSP_PLAN_VARIANT");
        if (v && *v) {  // (an EMPTY value is "not set": atoi("") would select variant 0 - it did, in two A/B sessions of round 5)
            const PlanInfo* var = adsp::variants_f32(&n);
            const int i = atoi(v);
            if (i >= 0 && i < n && var[i].M == M && var[i].FQ == FQ) return &var[i];
        }
    }
#endif
    const PlanInfo* tab = fmt == ADSP_FORMAT_S16_F64 ? adsp::plans_s16_f64(&n) : fmt == ADSP_FORMAT_S16 ? adsp::plans_s16(&n) : adsp::plans_f32(&n);
    for (int i = 0; i < n; ++i)
        if (tab[i].M == M && tab[i].FQ == FQ) return &tab[i];
    return nullptr;
}

const PlanInfo* find_plan_any_fn(int M, int fmt) {
    int n = 0;
    const PlanInfo* tab = fmt == ADSP_FORMAT_S16_F64 ? adsp::plans_s16_f64(&n) : fmt == ADSP_FORMAT_S16 ? adsp::plans_s16(&n) : adsp::plans_f32(&n);
    for (int i = 0; i < n; ++i)
        if (tab[i].M == M) return &tab[i];
    return nullptr;
}

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// Two kinds of geometry: "specialised" (chunk a power of two in 64..8192, F = 2N or 4N: chunk boundaries are
// compile-time constants in the kernel) and "generic" (any chunk divisible by 4, any supported power-of-two F).
// ... and, within the generic kind, "unaligned" chunks (not a multiple of 4, or shorter than 16 samples: the reference takes any
// chunk_size): float32 samples moved one dword at a time.
bool unaligned_chunk(int N) { return N % 4 != 0 || N < 16; }

int check_geometry(int N, int F, int fmt, const PlanInfo** out, bool* generic) {
#ifndef ADSP_TUNING_BUILD
    if (const char* v = getenv("ADSP_PLAN_VARIANT"))
        if (*v) return fail(ADSP_ERR_STATE, "ADSP_PLAN_VARIANT=%s is set, but this is the product library: the A/B plan variants (and the ablation / persistent-block "
                            "kernels) live in libadsp_tuning.so - `make -C pyaudiodsptools_amd/csrc tuning`, then ADSP_LIB=<path>/libadsp_tuning.so", v);
#endif
    if (fmt != ADSP_FORMAT_F32 && fmt != ADSP_FORMAT_S16 && fmt != ADSP_FORMAT_S16_F64)
        return fail(ADSP_ERR_ARG, "sample_format %d: need ADSP_FORMAT_F32, ADSP_FORMAT_S16 or ADSP_FORMAT_S16_F64", fmt);
    if (N < 4) return fail(ADSP_ERR_ARG, "chunk_size %d: need at least 4 samples", N);
    if (unaligned_chunk(N) && fmt != ADSP_FORMAT_F32)
        return fail(ADSP_ERR_ARG, "chunk_size %d: int16 engines need a multiple of 4, >= 16 (float32 engines take any chunk_size >= 4)", N);
    const bool three = F % 3 == 0 && is_pow2(F / 3);  // 3 * 2^k: the 1.5 N windows of the specialised kernels only
    if (!(is_pow2(F) || three) || F < 128 || F > 32768)
        return fail(ADSP_ERR_ARG, "fft_size %d: need a power of two in 128..32768 (or 1.5 x a power-of-two chunk that has a plan)", F);
    const bool special = is_pow2(N) && N >= 64 && N <= 8192 && (F == 2 * N || F == 4 * N || 2 * F == 3 * N);
    if (three && !special) return fail(ADSP_ERR_ARG, "fft_size %d = 3 * 2^k is only available as 1.5 x chunk_size", F);
    const PlanInfo* p = special ? find_plan(F / 2, 4 * F / N, fmt) : find_plan_any_fn(F / 2, fmt);
    if (!p) return fail(ADSP_ERR_ARG, "no kernel plan for %d complex points", F / 2);
    if (unaligned_chunk(N) && !p->launch_unaligned)
        return fail(ADSP_ERR_ARG, "chunk_size %d is not a multiple of 4 (or < 16): the plan selected for %d complex points has no dword-access kernel (a tuning "
                    "plan chosen with ADSP_PLAN_VARIANT, or an int16 / fused-effect table): use the default float32 plans for such chunk sizes", N, F / 2);
    if (out) *out = p;
    if (generic) *generic = !special;
    return ADSP_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
struct adsp_engine {
    adsp_config cfg;
    const PlanInfo* plan;
    int M, logN, block_outputs;
    int epi_op;       // fused output effect (ADSP_EFFECT_*), 0 = none
    int lfo_len;      // tremolo: LFO table length and the reference's buffer length (EffectTremolo.py:40-45)
    long long lfo_copy_len;
    int epi_phase;
    int epi_replay;
    const PlanInfo* plan_epi;  // twin of `plan` whose kernel applies the effect (nullptr: none available)
    bool epi_prepared;
    float epi_p[3];
    int accumulate;   // 0 overwrite the output, 1 add to it (partitioned FIRs, mix bus), 2 add and clip to [-1, 1]
    bool generic;  // generic-geometry kernel (chunk not a power of two / F not 2N or 4N)
    bool unaligned;  // ... its dword-access form (chunk not a multiple of 4, or < 16 samples)
    char* ring;    // [ring_slots][C][N] samples of cfg.sample_format
    int ring_pos;  // slot of the most recent chunk
    void* tw;     // real4 / real2 tables: float for the float kernels, double for ADSP_FORMAT_S16_F64 engines
    void* pair;
    void* pair0;
    bool f64() const { return cfg.sample_format == ADSP_FORMAT_S16_F64; }
    char* zeros;   // 4*chunk_size zero bytes
    bool have_spectrum;
    bool real_spec;  // every Im H == 0: the kernel takes the 3-real-constants-per-pair path
    std::vector<float> host_spec;  // the spectrum last set, interleaved (adsp_bcast_spectrum sends the root's)
    std::vector<double> host_spec64;  // ... when it was given in float64 (adsp_set_spectrum_f64): broadcast as it is
    float* d_spec;                 // 2 (M + 1) floats on the device: the buffer the RCCL broadcast runs on (lazily allocated)
    // stream-ordered table updates (adsp_set_spectrum_async): two pinned staging buffers, reused alternately
    char* pin_tab[2];
    size_t pin_tab_bytes;
    hipEvent_t ev_tab[2];
    bool tab_busy[2];
    int tab_slot;
    int kernel_reach;  // kernel taps at negative circular indices (adsp_set_kernel_reach); < 0 = unknown: load the whole window
    char* stage_in;
    char* stage_out;
    size_t stage_elems;  // capacity in samples
    // large host calls (round 5): the batch moves in slabs through double-buffered pinned staging - the H2D copy of slab i + 1 and the D2H
    // copy of slab i - 1 run on copy streams of their own beside the kernel of slab i (apply_host_pipelined)
    struct HostPipe {
        char* pin_in[2] = {nullptr, nullptr};
        char* pin_out[2] = {nullptr, nullptr};
        char* d_in[2] = {nullptr, nullptr};
        char* d_out[2] = {nullptr, nullptr};
        size_t slab_bytes = 0;
        hipStream_t s_in = nullptr, s_k = nullptr, s_out = nullptr;
        hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_k[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    } hp;
    // small host calls skip the staging copies: the kernel reads a pinned, device-mapped copy of the caller's input
    // and writes the result straight into pinned host memory (two input slots: the ring update of call k may still be
    // reading slot k % 2 while the caller fills the other)
    char* pin_in[2];
    char* pin_out;
    size_t pin_bytes;
    int pin_slot;
    bool pin_busy[2];
    hipEvent_t ev_pin[2];
    hipEvent_t ev_kernel;   // recorded right after the kernel when want_kernel_event is set
    bool want_kernel_event;
    bool timing;
    // zero-copy ring steps issued on more than one stream (adsp_apply_ring): per-step events order a step after the
    // producers of the history slots it reads (RAW) and a producer after the last readers of the slot it overwrites (WAR)
    struct RingStep {
        long long step = -1;
        hipStream_t stream = nullptr;
        hipEvent_t in = nullptr, out = nullptr;  // recorded just before / just after the step's kernel
    };
    std::vector<RingStep> ring_steps;
    long long step_no;        // index of the next zero-copy step
    bool multi_stream;        // a stream switch has been seen: events are recorded from then on
    bool have_last_stream;
    hipStream_t last_stream;
    hipEvent_t ev_join;       // everything enqueued on the old stream when the first switch was seen
    // resident ring launches (adsp_ring_produce_begin/_end, adsp_apply_ring_resident): the producer side publishes steps
    // through a device sequence word, a consumer launch covers many steps and its workgroups wait for theirs
    struct ResidentLaunch {
        long long first = 0;
        int n = 0;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        hipStream_t waited_by = nullptr;  // the producer stream that already waits for `done` (one wait per launch, not per slot)
        bool waited = false;
    };
    std::vector<ResidentLaunch> resident_launches;  // every launch that may still be running (entries are reused once their
                                                    // `done` event has fired: the table grows with the launches in flight)
    hipEvent_t ev_pub;        // recorded on the producer stream behind the most recent publication
    bool have_pub;
    bool resident_mode;
    unsigned* d_seq;          // [0] sequence word = number of steps published so far, [1] time-out flag; fine-grained device memory
    unsigned pub_count;       // host copy of the sequence word once every enqueued publication has executed
    int pub_pending;          // slots handed out by adsp_ring_produce_begin since the last publication
    int lead;                 // steps published but not yet handed to a consumer launch (negative: consumers launched ahead)
    bool seq_by_copy;         // hipStreamWriteValue32 is not available: publications are 4-byte copies from pinned memory
    unsigned* pin_seq;        // pinned source values of such copies (kSeqPinned of them, reused round-robin)
    unsigned long long resident_timeout_ticks;
    // pipelined ring steps (adsp_ring_set_pipeline): step k runs on the library's own stream k % depth, so consecutive launches
    // overlap (the next one fills the CUs the previous one is draining); the caller's stream carries the producers only
    int pipe_depth = 1;
    hipStream_t pipe_stream[2] = {nullptr, nullptr};
    hipEvent_t pipe_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    unsigned pipe_ev_next = 0;
    // live session (adsp_live_*): one persistent launch consumes ring steps as they are published
    struct Live {
        bool active = false;
        const adsp::LivePlanInfo* plan = nullptr;
        unsigned* d_words = nullptr;      // fine-grained device memory: [0] seq [1] done [2] stop [3] fail [4 .. 4 + ncg) progress
        size_t d_words_n = 0;
        unsigned* h_words = nullptr;      // pinned, device-mapped host memory.  Written by the HOST: [0] host_seq [2] host_stop; written by the
                                          // GPU, in a cache line of their own 512 bytes further on (kLiveGpuWords): [0] host_done [3..6] relay diagnostics
        unsigned* h_words_dev = nullptr;  // its device address
        unsigned published = 0;           // steps published to the session so far
        unsigned pending = 0;             // slots handed out by adsp_live_slot since the last publication
        unsigned max_steps = 0;
        int out_slots = 0;
        int ncg = 0;
        hipStream_t stream = nullptr;
        hipStream_t own_stream = nullptr;  // highest priority: a hardware queue of its own (see adsp_live_start)
        unsigned long long* trace = nullptr;  // ADSP_LIVE_TRACE: pinned, mapped; 64 steps x 8 stamps of workgroup 1
        // tables of a session plan that is not the engine's own (config 3 runs on 8 points per thread): rebuilt at every start
        void *own_tw = nullptr, *own_pair = nullptr, *own_pair0 = nullptr;
        size_t own_tw_bytes = 0, own_pair_bytes = 0, own_pair0_bytes = 0;
        int load_mode = 2;
        double timeout_ms = 1000.0;
        bool pipeline_owned = false;            // started by adsp_apply_ring in pipeline mode 3 (the library feeds and stops it)
        unsigned long long* d_out_table = nullptr;  // per-step output addresses (inside d_words), out_table_mask + 1 entries
        unsigned out_table_mask = 0;
    } live;
    hipStream_t copy_stream;  // ring update of multi-step launches runs beside the kernel
    hipEvent_t ev_in_ready, ev_copy_done;
    bool copy_pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;   // recorded, not yet read
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_ev;  // recycled event pairs
    size_t plane() const { return (size_t)cfg.n_channels * (size_t)cfg.chunk_size; }                 // samples per chunk batch
    size_t ssize() const { return cfg.sample_format == ADSP_FORMAT_F32 ? sizeof(float) : sizeof(short); }  // bytes per sample
    size_t plane_bytes() const { return plane() * ssize(); }
};

namespace {

int set_device(const adsp_engine* e) {
    HIP_TRY(hipSetDevice(e->cfg.device_id));
    return ADSP_OK;
}

// async = false: blocking copies (the caller has drained the device: nothing is reading the tables).
// async = true : the tables are staged in pinned memory and copied ON `stream`, i.e. after every launch already queued
//                there and before every later one - no device-wide synchronisation, the filter changes between two steps.
template <class R, class HT>
int upload_pairs_t(adsp_engine* e, const HT* H, hipStream_t stream, bool async) {
    using V = Vec<R>;
    using T2 = typename V::T2;
    using T4 = typename V::T4;
    const int M = e->M;
    // A real spectrum (zero-phase kernel) makes c1, c4 real and c2 imaginary: 3 floats per pair instead of 6.
    bool real_spec = true;
    for (int k = 0; k <= M && real_spec; ++k) real_spec = H[2 * k + 1] == (HT)0;
    if (getenv("ADSP_FORCE_COMPLEX")) real_spec = false;  // tuning: A/B the two spectrum stages on the same filter
    e->real_spec = real_spec;
    if constexpr (std::is_same<HT, float>::value) {
        if (e->host_spec.data() != H) e->host_spec.assign(H, H + 2 * (size_t)(M + 1));
        e->host_spec64.clear();
    } else {
        e->host_spec.clear();
        if (e->host_spec64.data() != H) e->host_spec64.assign(H, H + 2 * (size_t)(M + 1));
    }
    std::vector<T4> tab;
    std::vector<T2> tab0;
    build_pair_tables<R, HT>(*e->plan, M, H, real_spec, tab, tab0);
    const size_t b1 = tab.size() * sizeof(T4), b0 = tab0.size() * sizeof(T2);
    if (async) {
        if (e->pin_tab_bytes < b1 + b0) {  // first use (the table size of an engine never changes afterwards)
            for (int i = 0; i < 2; ++i) {
                if (!e->pin_tab[i]) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->pin_tab[i]), b1 + b0, hipHostMallocDefault));
                if (!e->ev_tab[i]) HIP_TRY(hipEventCreateWithFlags(&e->ev_tab[i], hipEventDisableTiming));
            }
            e->pin_tab_bytes = b1 + b0;  // only once both buffers and both events exist
        }
        const int b = e->tab_slot ^= 1;
        if (e->tab_busy[b]) HIP_TRY(hipEventSynchronize(e->ev_tab[b]));  // the copy two updates ago read this buffer
        memcpy(e->pin_tab[b], tab.data(), b1);
        memcpy(e->pin_tab[b] + b1, tab0.data(), b0);
        HIP_TRY(hipMemcpyAsync(e->pair, e->pin_tab[b], b1, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(e->pair0, e->pin_tab[b] + b1, b0, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipEventRecord(e->ev_tab[b], stream));
        e->tab_busy[b] = true;
    } else {
        // synchronous copies from pageable memory: safe to free the vectors on return
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(e->pair, tab.data(), b1, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->pair0, tab0.data(), b0, hipMemcpyHostToDevice));
    }
    e->have_spectrum = true;
    return ADSP_OK;
}

// float32 or float64 spectrum into a float or float64 engine (exactly one of H32 / H64 is given)
int upload_pairs(adsp_engine* e, const float* H32, hipStream_t stream, bool async = false, const double* H64 = nullptr) {
    if (e->f64()) return H64 ? upload_pairs_t<double, double>(e, H64, stream, async) : upload_pairs_t<double, float>(e, H32, stream, async);
    if (!H64) return upload_pairs_t<float, float>(e, H32, stream, async);
    std::vector<float> h(2 * (size_t)(e->M + 1));
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)H64[i];
    return upload_pairs_t<float, float>(e, h.data(), stream, async);
}

int launch(adsp_engine* e, const void* d_in, void* d_out, int n_steps, hipStream_t stream, bool resident = false) {
    const adsp_config& c = e->cfg;
    // same transform; the twin kernel has the output effect / mix bus compiled in (the plain generic kernel can add)
    const bool twin = e->epi_op != 0 || e->accumulate == 2 || (e->accumulate == 1 && !e->generic);
    const PlanInfo& pl = twin ? *e->plan_epi : *e->plan;
    adsp::KernelArgs a;
    a.ring = e->ring;
    a.in = d_in;
    a.out = d_out;
    a.tw = e->tw;
    a.pair = e->pair;
    a.pair0 = e->pair0;
    a.zeros = e->zeros;
    a.ring_pos = e->ring_pos;
    a.ring_slots = c.ring_slots;
    a.C = c.n_channels;
    a.n_steps = n_steps;
    a.V = ((n_steps == 1 || resident) && !e->generic) ? c.chunk_size : e->block_outputs;
    a.in_ring = resident ? 1 : 0;
    // every step already published: nothing will wait, so the launch may run in the multi-step order (a channel group's steps
    // are neighbours in the grid: their window overlap is an L2 hit); otherwise strictly step-major
    a.step_tile = (resident && e->lead >= n_steps) ? n_steps : 1;
    if (resident && a.step_tile > 1 && e->have_pub) {
        // `lead` counts publications ENQUEUED on the producer stream.  The tiled order lets later-step workgroups occupy CU
        // slots ahead of earlier ones, so nothing of this launch may start before those publications have executed: an
        // event wait on the consumer stream (free when they already have) instead of in-kernel waiting
        if (hipEventQuery(e->ev_pub) != hipSuccess) HIP_TRY(hipStreamWaitEvent(stream, e->ev_pub, 0));
        (void)hipGetLastError();  // (hipErrorNotReady from the query is not an error)
    }
    a.seq = resident ? e->d_seq : nullptr;
    a.seq_base = e->pub_count - (unsigned)e->lead;  // (wraps like the word itself)
    a.seq_fail = resident ? e->d_seq + 1 : nullptr;
    a.seq_timeout = e->resident_timeout_ticks;
    a.N = c.chunk_size;
    a.nh = c.history_chunks;
    a.inv_n = 1.0f / (float)c.chunk_size;
    a.accumulate = e->accumulate;
    a.real_spec = e->real_spec ? 1 : 0;
    a.epi_phase = e->epi_phase;
    a.epi_replay = e->epi_replay;
    a.epi_op = e->epi_op;
    a.epi_p0 = e->epi_p[0];
    a.epi_p1 = e->epi_p[1];
    a.epi_p2 = e->epi_p[2];
    const long long total = (long long)n_steps * c.chunk_size;
    if (total + 8LL * c.fft_size >= 0x7fffffffLL)  // the kernel indexes a channel's time axis with 32-bit ints
        return fail(ADSP_ERR_ARG, "n_steps %d x chunk %d is too long for one call; split it", n_steps, c.chunk_size);
    a.nblk = (int)((total + a.V - 1) / a.V);
    // window positions >= out_offset + V + reach feed discarded outputs only: whole register pairs (4T samples) beyond
    // them are not fetched
    a.win_pairs = pl.P / 2;
    if (e->kernel_reach >= 0) {
        const int need = c.out_offset + a.V + e->kernel_reach, seg = 4 * pl.T;
        const int pairs = (need + seg - 1) / seg;
        if (pairs < a.win_pairs) a.win_pairs = pairs;
    }
    a.lookback = c.lookback;
    a.j0 = c.out_offset;
    a.ncg = (c.n_channels + pl.CPB - 1) / pl.CPB;
    // non-temporal loads for the part of a window no other block reads; the head and the tail (F - V samples each, whole register pairs)
    // stay in L2 for the neighbouring blocks of the channel
    a.nt_lo = 0;
    a.nt_hi = pl.P / 2;
    if (pl.P >= 64 && !e->generic && n_steps > 1 && !(getenv("ADSP_NT_HYBRID") && atoi(getenv("ADSP_NT_HYBRID")) == 0)) {  // (ADSP_NT_HYBRID=0: tuning A/B)
        const int seg = 4 * pl.T, overlap = (c.fft_size - a.V + seg - 1) / seg;
        if (2 * overlap < pl.P / 2) {
            a.nt_lo = overlap;
            a.nt_hi = pl.P / 2 - overlap;
        }
    }
    a.blk_iters = 1;
    a.self = nullptr;
#ifdef ADSP_TUNING_BUILD
    if (getenv("ADSP_PERSIST_BUILD")) {  // tuning: a library whose kernels were built with -DADSP_PERSIST=1 (they read their arguments from a.self)
        const char* bi = getenv("ADSP_BLK_ITERS");
        a.blk_iters = (bi && !resident && !e->generic && atoi(bi) > 1) ? atoi(bi) : 1;
        static void* d_args[64] = {nullptr};  // a small ring of argument copies: launches in flight never share one
        static unsigned d_next = 0;
        void*& slot = d_args[d_next++ % 64];
        if (!slot) HIP_TRY(hipMalloc(&slot, sizeof a));
        a.self = slot;
    }
#endif
    const long long grid = (long long)((a.ncg + 7) / 8) * 8 *
                           (resident ? (a.nblk + a.step_tile - 1) / a.step_tile * a.step_tile : (a.nblk + a.blk_iters - 1) / a.blk_iters);  // resident: whole step tiles
    if (grid > 0x7fffffffLL) return fail(ADSP_ERR_ARG, "launch too large (%lld workgroups)", grid);
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    if (e->timing) {
        if (!e->free_ev.empty()) {
            ev = e->free_ev.back();
            e->free_ev.pop_back();
        } else {
            HIP_TRY(hipEventCreate(&ev.first));
            HIP_TRY(hipEventCreate(&ev.second));
        }
        HIP_TRY(hipEventRecord(ev.first, stream));
    }
    if (a.self) HIP_TRY(hipMemcpyAsync(const_cast<void*>(a.self), &a, sizeof a, hipMemcpyHostToDevice, stream));  // (tuning builds; `a` is copied by the call)
    HIP_TRY(e->unaligned ? pl.launch_unaligned(a, (int)grid, stream) : e->generic ? pl.launch_generic(a, (int)grid, stream) : pl.launch(a, (int)grid, stream));
    if (e->want_kernel_event) HIP_TRY(hipEventRecord(e->ev_kernel, stream));
    if (e->timing) {
        HIP_TRY(hipEventRecord(ev.second, stream));
        e->timed.push_back(ev);
    }
    return ADSP_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// A session started by the library itself (adsp_ring_set_pipeline(engine, 3): adsp_apply_ring rides a live session) is wound down by
// any call that needs the engine in its ordinary state; a session the caller started (adsp_live_start) is the caller's to stop.
namespace {
int live_pipe_release(adsp_engine* e);
int live_pipe_acquire(adsp_engine* e, void** d_slot);
int live_pipe_apply(adsp_engine* e, void* d_out, hipStream_t stream);
int live_pipe_check(adsp_engine* e);
}
#define ADSP_NOT_LIVE(e)                                                                                                  \
    do {                                                                                                                  \
        if ((e)->live.active) {                                                                                           \
            if (!(e)->live.pipeline_owned)                                                                                \
                return fail(ADSP_ERR_STATE, "a live session is running (adsp_live_start): call adsp_live_stop first");    \
            const int rc_live_ = live_pipe_release(e);                                                                    \
            if (rc_live_) return rc_live_;                                                                                \
        }                                                                                                                 \
    } while (0)

#define ADSP_NOT_RESIDENT(e)                                                                                              \
    ADSP_NOT_LIVE(e);                                                                                                     \
    if ((e)->resident_mode)                                                                                                \
        return fail(ADSP_ERR_STATE, "the ring is in resident mode (adsp_ring_produce_* / adsp_apply_ring_resident): call adsp_ring_reset_order first")

extern "C" {

int adsp_version(void) { return ADSP_ABI_VERSION; }

const char* adsp_build_info(void) {
#ifdef ADSP_TUNING_BUILD
    return "tuning";
#else
    return "product";
#endif
}

const char* adsp_last_error(void) { return g_last_error.c_str(); }

int adsp_device_count(int* count) {
    if (!count) return fail(ADSP_ERR_ARG, "count is NULL");
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess || n <= 0) {
        *count = 0;
        (void)hipGetLastError();
        return fail(ADSP_ERR_NO_DEVICE, "no HIP device: %s", hipGetErrorString(err));
    }
    *count = n;
    return ADSP_OK;
}

int adsp_plan_supported(int chunk_size, int fft_size) { return check_geometry(chunk_size, fft_size, ADSP_FORMAT_F32, nullptr, nullptr); }

int adsp_plan_describe(int chunk_size, int fft_size, int* complex_points, int* points_per_thread,
                       int* threads_per_transform, int* channels_per_workgroup, int* lds_bytes) {
    const PlanInfo* p = nullptr;
    int rc = check_geometry(chunk_size, fft_size, ADSP_FORMAT_F32, &p, nullptr);
    if (rc) return rc;
    if (complex_points) *complex_points = p->M;
    if (points_per_thread) *points_per_thread = p->P;
    if (threads_per_transform) *threads_per_transform = p->T;
    if (channels_per_workgroup) *channels_per_workgroup = p->CPB;
    if (lds_bytes) *lds_bytes = p->lds_bytes;
    return ADSP_OK;
}

int adsp_create(const adsp_config* cfg, adsp_engine** out_engine) {
    if (!cfg || !out_engine) return fail(ADSP_ERR_ARG, "NULL argument");
    *out_engine = nullptr;
    const PlanInfo* pl = nullptr;
    bool generic = false;
    int rc = check_geometry(cfg->chunk_size, cfg->fft_size, cfg->sample_format, &pl, &generic);
    if (rc) return rc;
    const int N = cfg->chunk_size, F = cfg->fft_size, T2 = 2 * pl->T;
    if (cfg->n_channels <= 0) return fail(ADSP_ERR_ARG, "n_channels must be positive");
    if (cfg->history_chunks < 1 || cfg->history_chunks > ADSP_MAX_HISTORY)
        return fail(ADSP_ERR_ARG, "history_chunks %d out of range 1..%d", cfg->history_chunks, ADSP_MAX_HISTORY);
    if (!generic) {
        // the specialised kernels resolve the window / kept-slice phase in QUARTER chunks (a 4-way switch on
        // (t & (N-1)) >> (log2 N - 2)) and store whole register pairs: everything on the time axis is a multiple of N/4
        // (whole registers for every plan, whole register pairs for the plans with 16-byte I/O)
        const int Q4 = N / 4;
        if (Q4 % T2) return fail(ADSP_ERR_STATE, "internal: N/4 = %d is not a multiple of %d", Q4, T2);
        if (cfg->lookback <= 0 || cfg->lookback > cfg->history_chunks * N || cfg->lookback % Q4)
            return fail(ADSP_ERR_ARG, "lookback %d must be in (0, history_chunks*N] and a multiple of N/4 = %d", cfg->lookback, Q4);
        if (cfg->out_offset < 0 || cfg->out_offset % Q4 || cfg->out_offset + N > F)
            return fail(ADSP_ERR_ARG, "out_offset %d must be a multiple of N/4 = %d with out_offset + N <= F", cfg->out_offset, Q4);
    } else {
        // generic geometry: 16-byte accesses need everything on the time axis to be a multiple of 4 samples; kept
        // ranges are whole register segments (2T samples)
        if (cfg->lookback <= 0 || cfg->lookback > cfg->history_chunks * N || (cfg->lookback % 4 && !unaligned_chunk(N)))
            return fail(ADSP_ERR_ARG, "lookback %d must be in (0, history_chunks*N] and a multiple of 4", cfg->lookback);
        if (cfg->out_offset < 0 || cfg->out_offset % (2 * T2) || cfg->out_offset + 2 * T2 > F)
            return fail(ADSP_ERR_ARG, "out_offset %d must be a multiple of %d with out_offset + %d <= F", cfg->out_offset, 2 * T2, 2 * T2);
    }
    // kept sample i sits at input-time o - lookback + out_offset + i; it may not lie beyond the newest chunk
    if (cfg->out_offset > cfg->lookback)
        return fail(ADSP_ERR_ARG, "out_offset %d > lookback %d: kept samples would need future input", cfg->out_offset, cfg->lookback);
    int slots = cfg->ring_slots == 0 ? 2 * cfg->history_chunks : cfg->ring_slots;
    if (slots < 2) slots = 2;
    if (slots < cfg->history_chunks + 1) return fail(ADSP_ERR_ARG, "ring_slots must be >= history_chunks + 1");

    int ndev = 0;
    rc = adsp_device_count(&ndev);
    if (rc) return rc;
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(ADSP_ERR_ARG, "device_id %d out of range (%d devices)", cfg->device_id, ndev);

    adsp_engine* e = new adsp_engine();
    e->cfg = *cfg;
    e->cfg.ring_slots = slots;
    e->plan = pl;
    e->M = F / 2;
    e->logN = ilog2(N);
    e->generic = generic;
    e->unaligned = generic && unaligned_chunk(N);
    e->accumulate = 0;
    e->epi_op = 0;
    e->lfo_len = 0;
    e->lfo_copy_len = 0;
    e->epi_phase = 0;
    e->epi_replay = 0;
    e->plan_epi = nullptr;
    e->epi_prepared = false;
    if (e->cfg.sample_format == ADSP_FORMAT_F32) {
        int n = 0, ne = 0;
        const PlanInfo* tab = adsp::plans_f32(&n);
        const PlanInfo* tab_epi = adsp::plans_f32_epi(&ne);
        if (pl >= tab && pl < tab + n && ne == n) e->plan_epi = &tab_epi[pl - tab];
    }
    e->epi_p[0] = e->epi_p[1] = e->epi_p[2] = 0.f;
    // samples kept per transform: one chunk for the specialised kernels' single-step launches; the generic kernel
    // always tiles the time axis with block_outputs (default: as many whole segments as the transform offers)
    e->block_outputs = generic ? (F - cfg->out_offset) / (2 * T2) * (2 * T2) : N;
    e->ring = nullptr;
    e->ring_pos = slots - 1;
    e->tw = nullptr;
    e->pair = nullptr;
    e->pair0 = nullptr;
    e->zeros = nullptr;
    e->have_spectrum = false;
    e->real_spec = false;
    e->d_spec = nullptr;
    e->kernel_reach = -1;
    e->pin_tab[0] = e->pin_tab[1] = nullptr;
    e->pin_tab_bytes = 0;
    e->ev_tab[0] = e->ev_tab[1] = nullptr;
    e->tab_busy[0] = e->tab_busy[1] = false;
    e->tab_slot = 0;
    e->stage_in = e->stage_out = nullptr;
    e->stage_elems = 0;
    e->pin_in[0] = e->pin_in[1] = e->pin_out = nullptr;
    e->pin_bytes = 0;
    e->pin_slot = 0;
    e->pin_busy[0] = e->pin_busy[1] = false;
    e->ev_pin[0] = e->ev_pin[1] = e->ev_kernel = nullptr;
    e->want_kernel_event = false;
    e->timing = false;
    e->ev_pub = nullptr;
    e->have_pub = false;
    e->resident_mode = false;
    e->d_seq = nullptr;
    e->pub_count = 0;
    e->pub_pending = 0;
    e->lead = 0;
    e->seq_by_copy = getenv("ADSP_SEQ_COPY") != nullptr;  // tuning: publish through 4-byte copies instead of hipStreamWriteValue32
    e->pin_seq = nullptr;
    e->resident_timeout_ticks = 25000000ull;  // 250 ms of the 100 MHz clock
    e->step_no = 0;
    e->multi_stream = e->have_last_stream = false;
    e->last_stream = nullptr;
    e->ev_join = nullptr;
    e->copy_stream = nullptr;
    e->ev_in_ready = e->ev_copy_done = nullptr;
    e->copy_pending = false;

    auto bail = [&](int code) {
        adsp_destroy(e);
        return code;
    };
    if ((rc = set_device(e))) return bail(rc);
    hipError_t err;
    if ((err = (e->unaligned ? pl->prepare_unaligned() : generic ? pl->prepare_generic() : pl->prepare())) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(err)));
    const size_t ring_bytes = (size_t)slots * e->plane_bytes();
    if ((err = hipMalloc(&e->ring, ring_bytes)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMalloc ring (%zu bytes): %s", ring_bytes, hipGetErrorString(err)));
    if ((err = hipMemset(e->ring, 0, ring_bytes)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMemset: %s", hipGetErrorString(err)));
    std::vector<float4> tw;
    std::vector<double4> tw64;
    if (e->f64()) build_twiddles<double>(*pl, tw64); else build_twiddles<float>(*pl, tw);
    const size_t tw_n = e->f64() ? tw64.size() : tw.size(), t4 = e->f64() ? sizeof(double4) : sizeof(float4);
    if ((int)tw_n != pl->tw_total) return bail(fail(ADSP_ERR_STATE, "internal: twiddle count %zu != %d", tw_n, pl->tw_total));
    if ((err = hipMalloc(&e->tw, (tw_n + 1) * t4)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMalloc: %s", hipGetErrorString(err)));
    if (tw_n && (err = hipMemcpy(e->tw, e->f64() ? (const void*)tw64.data() : (const void*)tw.data(), tw_n * t4, hipMemcpyHostToDevice)) != hipSuccess)
        return bail(fail(ADSP_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(err)));
    const int R = pl->rad[pl->NP - 1];  // radix of the paired passes
    if ((err = hipMalloc(&e->pair, (size_t)(pl->P / 2) * 3 * pl->T * t4)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMalloc: %s", hipGetErrorString(err)));
    if ((err = hipMalloc(&e->pair0, (size_t)(R + 1) * 3 * (t4 / 2))) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMalloc: %s", hipGetErrorString(err)));
    if ((err = hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(err)));
    if ((err = hipEventCreateWithFlags(&e->ev_in_ready, hipEventDisableTiming)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(err)));
    if ((err = hipEventCreateWithFlags(&e->ev_copy_done, hipEventDisableTiming)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(err)));
    if ((err = hipMalloc(&e->zeros, (size_t)N * sizeof(float))) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMalloc: %s", hipGetErrorString(err)));
    if ((err = hipMemset(e->zeros, 0, (size_t)N * sizeof(float))) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMemset: %s", hipGetErrorString(err)));
    *out_engine = e;
    return ADSP_OK;
}

int adsp_destroy(adsp_engine* e) {
    if (!e) return ADSP_OK;
    (void)hipSetDevice(e->cfg.device_id);
    if (e->live.h_words) __atomic_store_n(e->live.h_words + 2, 1u, __ATOMIC_RELEASE);  // a session still running ends at its next poll
    (void)hipDeviceSynchronize();
    if (e->ring) (void)hipFree(e->ring);
    if (e->tw) (void)hipFree(e->tw);
    if (e->pair) (void)hipFree(e->pair);
    if (e->pair0) (void)hipFree(e->pair0);
    if (e->zeros) (void)hipFree(e->zeros);
    if (e->d_spec) (void)hipFree(e->d_spec);
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->ev_in_ready) (void)hipEventDestroy(e->ev_in_ready);
    if (e->ev_copy_done) (void)hipEventDestroy(e->ev_copy_done);
    if (e->stage_in) (void)hipFree(e->stage_in);
    if (e->stage_out) (void)hipFree(e->stage_out);
    for (char* p : {e->hp.d_in[0], e->hp.d_in[1], e->hp.d_out[0], e->hp.d_out[1]})
        if (p) (void)hipFree(p);
    for (char* p : {e->hp.pin_in[0], e->hp.pin_in[1], e->hp.pin_out[0], e->hp.pin_out[1]})
        if (p) (void)hipHostFree(p);
    for (hipStream_t st : {e->hp.s_in, e->hp.s_k, e->hp.s_out})
        if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : {e->hp.ev_in[0], e->hp.ev_in[1], e->hp.ev_k[0], e->hp.ev_k[1], e->hp.ev_out[0], e->hp.ev_out[1]})
        if (ev) (void)hipEventDestroy(ev);
    for (char* p : {e->pin_in[0], e->pin_in[1], e->pin_out, e->pin_tab[0], e->pin_tab[1]})
        if (p) (void)hipHostFree(p);
    for (hipEvent_t ev : {e->ev_tab[0], e->ev_tab[1]})
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : {e->ev_pin[0], e->ev_pin[1], e->ev_kernel})
        if (ev) (void)hipEventDestroy(ev);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    for (auto& rl : e->resident_launches)
        if (rl.done) (void)hipEventDestroy(rl.done);
    if (e->ev_pub) (void)hipEventDestroy(e->ev_pub);
    for (hipStream_t st : e->pipe_stream)
        if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : e->pipe_ev)
        if (ev) (void)hipEventDestroy(ev);
    if (e->d_seq) (void)hipFree(e->d_seq);
    if (e->pin_seq) (void)hipHostFree(e->pin_seq);
    if (e->live.h_words) (void)hipHostFree(e->live.h_words);
    if (e->live.d_words) (void)hipFree(e->live.d_words);
    if (e->live.own_stream) (void)hipStreamDestroy(e->live.own_stream);
    if (e->live.trace) (void)hipHostFree(e->live.trace);
    for (void* p : {e->live.own_tw, e->live.own_pair, e->live.own_pair0})
        if (p) (void)hipFree(p);
    for (auto& st : e->ring_steps) {
        if (st.in) (void)hipEventDestroy(st.in);
        if (st.out) (void)hipEventDestroy(st.out);
    }
    for (auto& v : {&e->timed, &e->free_ev})
        for (auto& p : *v) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    delete e;
    return ADSP_OK;
}

int adsp_set_spectrum(adsp_engine* e, const float* spectrum, int n_bins) {
    if (!e || !spectrum) return fail(ADSP_ERR_ARG, "NULL argument");
    ADSP_NOT_LIVE(e);
    if (n_bins != e->M + 1) return fail(ADSP_ERR_ARG, "n_bins %d != fft_size/2+1 = %d", n_bins, e->M + 1);
    int rc = set_device(e);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());  // the tables may still be in use by queued launches
    e->kernel_reach = -1;  // the hint described the previous kernel
    return upload_pairs(e, spectrum, nullptr);
}

int adsp_set_spectrum_f64(adsp_engine* e, const double* spectrum, int n_bins) {
    if (!e || !spectrum) return fail(ADSP_ERR_ARG, "NULL argument");
    ADSP_NOT_LIVE(e);
    if (n_bins != e->M + 1) return fail(ADSP_ERR_ARG, "n_bins %d != fft_size/2+1 = %d", n_bins, e->M + 1);
    int rc = set_device(e);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    e->kernel_reach = -1;
    return upload_pairs(e, nullptr, nullptr, false, spectrum);
}

int adsp_set_spectrum_async(adsp_engine* e, const float* spectrum, int n_bins, void* stream) {
    if (!e || !spectrum) return fail(ADSP_ERR_ARG, "NULL argument");
    ADSP_NOT_LIVE(e);
    if (n_bins != e->M + 1) return fail(ADSP_ERR_ARG, "n_bins %d != fft_size/2+1 = %d", n_bins, e->M + 1);
    int rc = set_device(e);
    if (rc) return rc;
    e->kernel_reach = -1;
    return upload_pairs(e, spectrum, (hipStream_t)stream, true);
}

int adsp_set_spectrum_device(adsp_engine* e, const float* d_spectrum, int n_bins, void* stream) {
    if (!e || !d_spectrum) return fail(ADSP_ERR_ARG, "NULL argument");
    ADSP_NOT_LIVE(e);
    if (n_bins != e->M + 1) return fail(ADSP_ERR_ARG, "n_bins %d != fft_size/2+1 = %d", n_bins, e->M + 1);
    int rc = set_device(e);
    if (rc) return rc;
    // the pair tables are float64 host arithmetic: fetch the spectrum ON `stream` (ordered after whatever produced it,
    // e.g. an RCCL broadcast enqueued there), wait for that stream only, then update the tables stream-ordered
    std::vector<float> host((size_t)2 * n_bins);
    HIP_TRY(hipMemcpyAsync(host.data(), d_spectrum, host.size() * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    e->kernel_reach = -1;
    return upload_pairs(e, host.data(), (hipStream_t)stream, true);
}

namespace {
// What travels: the root's spectrum as it was given - 2 (M + 1) floats, or 2 (M + 1) doubles moved as twice as many floats
// (a broadcast moves bytes) for a float64 spectrum (adsp_set_spectrum_f64: the exact-FFT engines keep their precision).
size_t bcast_floats(const adsp_engine* e, bool is64) { return 2 * (size_t)(e->M + 1) * (is64 ? 2 : 1); }

int bcast_stage_root(adsp_engine* e, bool is64) {  // root: spectrum -> its device buffer, on its side stream
    const size_t bytes = bcast_floats(e, is64) * sizeof(float);
    HIP_TRY(hipMemcpyAsync(e->d_spec, is64 ? (const void*)e->host_spec64.data() : (const void*)e->host_spec.data(), bytes,
                           hipMemcpyHostToDevice, e->copy_stream));
    return ADSP_OK;
}

int bcast_buffer(adsp_engine* e) {  // large enough for either precision
    if (!e->d_spec) HIP_TRY(hipMalloc(&e->d_spec, bcast_floats(e, true) * sizeof(float)));
    return ADSP_OK;
}

// every engine, the root included, rebuilds its tables from what the collective left in ITS memory
int bcast_adopt(adsp_engine* e, bool is64, int reach) {
    int rc = set_device(e);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());  // set-up path: the tables may still be in use by queued launches
    if (is64) {
        std::vector<double> host(2 * (size_t)(e->M + 1));
        HIP_TRY(hipMemcpy(host.data(), e->d_spec, host.size() * sizeof(double), hipMemcpyDeviceToHost));
        if ((rc = upload_pairs(e, nullptr, nullptr, false, host.data()))) return rc;
    } else {
        if ((rc = adsp_set_spectrum_device(e, e->d_spec, e->M + 1, e->copy_stream))) return rc;
        HIP_TRY(hipStreamSynchronize(e->copy_stream));
    }
    e->kernel_reach = reach;  // same kernel, same reach
    return ADSP_OK;
}
}  // namespace

// The one collective of the multi-GPU path: every engine takes over the ROOT engine's filter.  One process, n devices.
int adsp_bcast_spectrum(adsp_engine* const* engines, int n, int root) {
    if (!engines || n < 1) return fail(ADSP_ERR_ARG, "need at least one engine");
    if (root < 0 || root >= n) return fail(ADSP_ERR_ARG, "root %d out of range 0..%d", root, n - 1);
    for (int i = 0; i < n; ++i) {
        if (!engines[i]) return fail(ADSP_ERR_ARG, "engine %d is NULL", i);
        for (int j = 0; j < i; ++j)
            if (engines[j] == engines[i] || engines[j]->cfg.device_id == engines[i]->cfg.device_id)
                return fail(ADSP_ERR_ARG, "engines %d and %d share device %d: one engine per GPU (RCCL ranks are devices)", j, i,
                            engines[i]->cfg.device_id);
    }
    const adsp_engine* r = engines[root];
    if (!r->have_spectrum) return fail(ADSP_ERR_STATE, "the root engine has no spectrum yet (adsp_set_spectrum)");
    const bool is64 = r->host_spec.empty();
    if (is64 && r->host_spec64.empty()) return fail(ADSP_ERR_STATE, "internal: the root engine kept no copy of its spectrum");
    for (int i = 0; i < n; ++i) {
        // a spectrum only means something together with the window geometry it was designed for
        const adsp_config &a = engines[i]->cfg, &b = r->cfg;
        if (a.chunk_size != b.chunk_size || a.fft_size != b.fft_size || a.history_chunks != b.history_chunks ||
            a.lookback != b.lookback || a.out_offset != b.out_offset || a.sample_format != b.sample_format)
            return fail(ADSP_ERR_ARG, "engine %d has a different geometry than the root engine (chunk %d/%d, fft %d/%d, lookback %d/%d, "
                        "out_offset %d/%d)", i, a.chunk_size, b.chunk_size, a.fft_size, b.fft_size, a.lookback, b.lookback, a.out_offset, b.out_offset);
    }
    const size_t count = bcast_floats(r, is64);
    std::vector<float*> bufs(n);
    std::vector<int> devs(n);
    std::vector<hipStream_t> streams(n);
    for (int i = 0; i < n; ++i) {
        adsp_engine* e = engines[i];
        int rc = set_device(e);
        if (rc) return rc;
        if ((rc = bcast_buffer(e))) return rc;
        bufs[i] = e->d_spec;
        devs[i] = e->cfg.device_id;
        streams[i] = e->copy_stream;  // the engine's own side stream: nothing of the caller's is ordered behind the collective
    }
    int rc = set_device(engines[root]);
    if (rc) return rc;
    if ((rc = bcast_stage_root(engines[root], is64))) return rc;
    if ((rc = adsp::rccl_broadcast(bufs.data(), devs.data(), streams.data(), n, count, root))) return rc;
    const int reach = r->kernel_reach;
    for (int i = 0; i < n; ++i)
        if ((rc = bcast_adopt(engines[i], is64, reach))) return rc;
    return ADSP_OK;
}

int adsp_rccl_unique_id(char* unique_id) {
    if (!unique_id) return fail(ADSP_ERR_ARG, "unique_id is NULL");
    return adsp::rccl_unique_id(unique_id);
}

// The same collective for a ONE-PROCESS-PER-GPU job: this process holds rank `rank` of `world`.
// Both collectives (header, spectrum) are entered by EVERY rank whatever a rank finds wrong with its own engine - a rank that
// returned early would leave the others blocked inside RCCL: the root reports its own trouble IN the header (everyone then skips the
// spectrum and fails), a rank whose engine does not match the header still receives the spectrum (into a scratch buffer of the
// root's size) and fails afterwards.  Only a failure of RCCL itself cannot be made symmetric: abort the job then.
int adsp_bcast_spectrum_rank(adsp_engine* e, const char* unique_id, int rank, int world, int root) {
    if (!e || !unique_id) return fail(ADSP_ERR_ARG, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world || root < 0 || root >= world)
        return fail(ADSP_ERR_ARG, "rank %d / root %d out of range for a world of %d", rank, root, world);
    int rc = set_device(e);
    if (rc) return rc;
    if ((rc = bcast_buffer(e))) return rc;
    // header first: the root says what it sends, every rank checks it against its own engine (a rank that derived another
    // window layout must fail loudly instead of filtering with the wrong offsets)
    constexpr int kHdr = 12;
    float* d_hdr = e->d_spec;  // the spectrum buffer doubles as the header buffer
    float hdr[kHdr] = {0};
    bool is64 = false;
    int root_trouble = 0;  // 1: no spectrum yet, 2: no host copy of it
    if (rank == root) {
        if (!e->have_spectrum) root_trouble = 1;
        is64 = e->host_spec.empty();
        if (!root_trouble && is64 && e->host_spec64.empty()) root_trouble = 2;
        const float h[kHdr] = {(float)e->cfg.chunk_size, (float)e->cfg.fft_size, (float)e->cfg.history_chunks, (float)e->cfg.lookback,
                               (float)e->cfg.out_offset, (float)e->cfg.sample_format, is64 ? 1.f : 0.f, (float)e->kernel_reach, (float)root_trouble, 0.f, 0.f, 0.f};
        memcpy(hdr, h, sizeof hdr);
        HIP_TRY(hipMemcpyAsync(d_hdr, hdr, sizeof hdr, hipMemcpyHostToDevice, e->copy_stream));
        HIP_TRY(hipStreamSynchronize(e->copy_stream));  // (hdr is a stack array)
    }
    if ((rc = adsp::rccl_broadcast_rank(unique_id, rank, world, root, e->cfg.device_id, d_hdr, kHdr, e->copy_stream))) return rc;
    HIP_TRY(hipMemcpyAsync(hdr, d_hdr, sizeof hdr, hipMemcpyDeviceToHost, e->copy_stream));
    HIP_TRY(hipStreamSynchronize(e->copy_stream));
    if ((int)hdr[8] != 0)  // every rank reads the same header: every rank leaves here, nobody enters the second collective
        return fail(ADSP_ERR_STATE, (int)hdr[8] == 1 ? "rank %d (the root) has no spectrum yet (adsp_set_spectrum): nothing was broadcast"
                                                     : "internal: rank %d (the root) kept no copy of its spectrum: nothing was broadcast", root);
    const adsp_config& c = e->cfg;
    const int mine[6] = {c.chunk_size, c.fft_size, c.history_chunks, c.lookback, c.out_offset, c.sample_format};
    bool match = true;
    for (int i = 0; i < 6; ++i) match = match && (int)hdr[i] == mine[i];
    is64 = hdr[6] != 0.f;
    const int reach = (int)hdr[7];
    // what the root sends: 2 (F/2 + 1) floats of ITS transform length (twice as many for a float64 spectrum)
    const size_t count = 2 * (size_t)((int)hdr[1] / 2 + 1) * (is64 ? 2 : 1);
    if (!match) {
        float* scratch = nullptr;
        HIP_TRY(hipMalloc(&scratch, count * sizeof(float)));
        rc = adsp::rccl_broadcast_rank(unique_id, rank, world, root, e->cfg.device_id, scratch, count, e->copy_stream);
        (void)hipStreamSynchronize(e->copy_stream);
        (void)hipFree(scratch);
        if (rc) return rc;
        return fail(ADSP_ERR_ARG, "rank %d: engine geometry (chunk %d, fft %d, history %d, lookback %d, out_offset %d, format %d) differs from "
                    "rank %d's (%d, %d, %d, %d, %d, %d); the spectrum was received and dropped, this engine keeps its own filter", rank, mine[0], mine[1],
                    mine[2], mine[3], mine[4], mine[5], root, (int)hdr[0], (int)hdr[1], (int)hdr[2], (int)hdr[3], (int)hdr[4], (int)hdr[5]);
    }
    if (rank == root && (rc = bcast_stage_root(e, is64))) return rc;
    if ((rc = adsp::rccl_broadcast_rank(unique_id, rank, world, root, e->cfg.device_id, e->d_spec, count, e->copy_stream))) return rc;
    return bcast_adopt(e, is64, reach);
}

int adsp_rccl_finalize(void) { return adsp::rccl_finalize(); }

int adsp_get_spectrum(const adsp_engine* e, float* spectrum, int n_bins) {
    if (!e || !spectrum) return fail(ADSP_ERR_ARG, "NULL argument");
    if (n_bins != e->M + 1) return fail(ADSP_ERR_ARG, "n_bins %d != fft_size/2+1 = %d", n_bins, e->M + 1);
    if (!e->have_spectrum) return fail(ADSP_ERR_STATE, "adsp_set_spectrum has not been called");
    const size_t n = 2 * (size_t)n_bins;
    if (!e->host_spec.empty()) memcpy(spectrum, e->host_spec.data(), n * sizeof(float));
    else for (size_t i = 0; i < n; ++i) spectrum[i] = (float)e->host_spec64[i];
    return ADSP_OK;
}

int adsp_rccl_version(int* version) {
    if (!version) return fail(ADSP_ERR_ARG, "version is NULL");
    return adsp::rccl_version(version);
}

int adsp_spectrum_is_real(const adsp_engine* e, int* is_real) {
    if (!e || !is_real) return fail(ADSP_ERR_ARG, "NULL argument");
    if (!e->have_spectrum) return fail(ADSP_ERR_STATE, "adsp_set_spectrum has not been called");
    *is_real = e->real_spec ? 1 : 0;
    return ADSP_OK;
}

int adsp_set_kernel_reach(adsp_engine* e, int taps_at_negative_indices) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    if (taps_at_negative_indices >= e->cfg.fft_size) return fail(ADSP_ERR_ARG, "kernel reach %d >= fft_size", taps_at_negative_indices);
    e->kernel_reach = taps_at_negative_indices < 0 ? -1 : taps_at_negative_indices;
    return ADSP_OK;
}

int adsp_set_block_outputs(adsp_engine* e, int v) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    ADSP_NOT_LIVE(e);
    // generic kernel: whole register pairs; specialised kernels: quarter chunks (their store phase is a 4-way switch)
    const int T2 = e->generic ? 4 * e->plan->T : e->cfg.chunk_size / 4;
    if (v <= 0 || v % T2 || e->cfg.out_offset + v > e->cfg.fft_size)
        return fail(ADSP_ERR_ARG, "block_outputs %d must be a positive multiple of %d with out_offset + block_outputs <= fft_size", v, T2);
    // the window must not need input newer than what a block's last output may see:
    // newest input used by the block = o - lookback + F - 1 may exceed the data, that part is zero-filled and
    // only feeds discarded circular positions as long as out_offset + V <= F (checked above).
    e->block_outputs = v;
    return ADSP_OK;
}

namespace {
int prepare_twin(adsp_engine* e) {
    if (e->unaligned) return fail(ADSP_ERR_ARG, "chunk_size %d is not a multiple of 4 (or < 16): fused effects and the clipping mix bus need an aligned chunk size - "
                                  "run the effect as its own pass (adsp_effect_device)", e->cfg.chunk_size);
    if (!e->plan_epi) return fail(ADSP_ERR_STATE, "no effect kernel for this (tuning) plan");
    if (!e->epi_prepared) {
        int rc = set_device(e);
        if (rc) return rc;
        HIP_TRY(e->generic ? e->plan_epi->prepare_generic() : e->plan_epi->prepare());
        e->epi_prepared = true;
    }
    return ADSP_OK;
}

// The reference's tremolo keeps a buffer of LFO tables and cuts each chunk off its front (EffectTremolo.py:40-45).
// Its length is the whole state: the buffer always ends on a table end, so the next chunk starts at table index
// (-length) mod table.  One quirk is kept: when the buffer holds EXACTLY one chunk, `copy[-0:]` keeps all of it, and
// every later chunk replays that same segment.  Returns how many of the next max_steps chunks run on contiguously.
int tremolo_run(adsp_engine* e, int max_steps, int* phase) {
    const long long N = e->cfg.chunk_size, L = e->lfo_len;
    long long len = e->lfo_copy_len;
    e->epi_replay = 0;
    if (len == N) {  // the buffer is stuck on one chunk's worth of table: this and every later chunk replay it
        e->epi_replay = 1;
        *phase = (int)((L - N % L) % L);
        return max_steps;
    }
    while (len < N) len += L;
    *phase = (int)((L - len % L) % L);
    int run = 0;
    while (run < max_steps) {
        while (len < N) len += L;
        ++run;
        if (len == N) break;  // replayed from now on: the next run starts at the same table index again
        len -= N;
    }
    e->lfo_copy_len = len;
    return run;
}
}  // namespace

int adsp_set_epilogue(adsp_engine* e, int effect, float p0, float p1, float p2) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    ADSP_NOT_LIVE(e);
    if (effect < ADSP_EFFECT_NONE || effect > ADSP_EFFECT_BIT_CRUSHER) return fail(ADSP_ERR_ARG, "unknown effect %d", effect);
    if (effect != ADSP_EFFECT_NONE && e->cfg.sample_format != ADSP_FORMAT_F32)
        return fail(ADSP_ERR_ARG, "fused effects need a float32 engine");
    if (effect == ADSP_EFFECT_TREMOLO && !(p2 >= 1.f && p2 <= 8388608.f && p2 == (float)(int)p2))
        return fail(ADSP_ERR_ARG, "tremolo: p2 must be the LFO table length, an integer in 1..2^23");
    if (effect != ADSP_EFFECT_NONE) {
        int rc = prepare_twin(e);
        if (rc) return rc;
    }
    e->epi_op = effect;
    e->epi_p[0] = p0;
    e->epi_p[1] = p1;
    e->epi_p[2] = p2;
    e->lfo_len = effect == ADSP_EFFECT_TREMOLO ? (int)p2 : 0;
    e->lfo_copy_len = e->lfo_len;  // a fresh LFO: one table in the buffer (EffectTremolo.py:24)
    e->epi_phase = 0;
    e->epi_replay = 0;
    return ADSP_OK;
}

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

int adsp_get_epilogue_state(const adsp_engine* e, long long* state) {
    if (!e || !state) return fail(ADSP_ERR_ARG, "NULL argument");
    *state = e->lfo_copy_len;  // the fused tremolo's whole state: the length of the reference's LFO buffer
    return ADSP_OK;
}

int adsp_set_epilogue_state(adsp_engine* e, long long state) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    if (e->epi_op != ADSP_EFFECT_TREMOLO) return state == 0 ? ADSP_OK : fail(ADSP_ERR_STATE, "only a fused tremolo carries state");
    if (state < 1 || state > (long long)e->lfo_len + e->cfg.chunk_size)
        return fail(ADSP_ERR_ARG, "tremolo state %lld out of range 1..%lld", state, (long long)e->lfo_len + e->cfg.chunk_size);
    e->lfo_copy_len = state;
    return ADSP_OK;
}

int adsp_set_accumulate(adsp_engine* e, int mode) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    ADSP_NOT_LIVE(e);
    if (mode < 0 || mode > 2) return fail(ADSP_ERR_ARG, "accumulate mode must be 0, 1 or 2");
    if (mode && e->cfg.sample_format != ADSP_FORMAT_F32) return fail(ADSP_ERR_ARG, "accumulating output needs a float32 engine");
    if (mode == 2 || (mode == 1 && !e->generic)) {  // these run on the twin kernel
        int rc = prepare_twin(e);
        if (rc) return rc;
    }
    e->accumulate = mode;
    return ADSP_OK;
}

namespace {
// ---- stream ordering of zero-copy ring steps ------------------------------------------------------------------
// Step k reads ring slots k - history .. k (written by the producers of those steps) and the producer of step k
// overwrites the slot of step k - ring_slots, which steps k - ring_slots .. k - ring_slots + history have read.  On ONE
// stream the stream orders all of it and nothing is recorded.  The first time a step arrives on a different stream the
// new stream joins the old one once (ev_join); from then on every step records an event before and after its kernel and
// a step / producer on stream s waits for exactly the events of the conflicting steps that ran on other streams.
void ring_forget_steps(adsp_engine* e) {  // after a device-wide synchronisation: nothing is in flight
    for (auto& st : e->ring_steps) st.step = -1;
    e->multi_stream = false;
    e->have_last_stream = false;
    for (auto& rl : e->resident_launches) rl.n = 0;
    e->resident_mode = false;
}

constexpr int kSeqPinned = 4096;

int resident_prepare(adsp_engine* e) {
    if (e->generic) return fail(ADSP_ERR_STATE, "resident ring launches need a specialised kernel (power-of-two chunk, F = 1.5 / 2 / 4 N)");
    if (e->multi_stream) return fail(ADSP_ERR_STATE, "ring steps are in flight on several streams: call adsp_ring_reset_order before the resident calls");
    if (!e->d_seq) {
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, 2 * sizeof(unsigned), hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            HIP_TRY(hipMalloc(&p, 2 * sizeof(unsigned)));
        }
        e->d_seq = static_cast<unsigned*>(p);
        HIP_TRY(hipMemset(e->d_seq, 0, 2 * sizeof(unsigned)));
        HIP_TRY(hipDeviceSynchronize());
        e->pub_count = 0;
        e->resident_launches.reserve(8);
    }
    e->resident_mode = true;
    return ADSP_OK;
}

int ring_enter_multi_stream(adsp_engine* e, hipStream_t stream) {
    if (!e->ev_join) HIP_TRY(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(e->ev_join, e->last_stream));
    HIP_TRY(hipStreamWaitEvent(stream, e->ev_join, 0));
    if (e->ring_steps.empty()) e->ring_steps.resize((size_t)e->cfg.ring_slots + e->cfg.history_chunks + 2);
    for (auto& st : e->ring_steps) st.step = -1;
    e->multi_stream = true;
    return ADSP_OK;
}

// make `stream` wait for step `k`'s event (`out`: its kernel has finished; otherwise: its input was complete)
int ring_wait_step(adsp_engine* e, long long k, hipStream_t stream, bool out) {
    if (k < 0) return ADSP_OK;
    const auto& st = e->ring_steps[(size_t)(k % (long long)e->ring_steps.size())];
    if (st.step != k) {  // issued before the first stream switch: covered by the join event
        HIP_TRY(hipStreamWaitEvent(stream, e->ev_join, 0));
        return ADSP_OK;
    }
    if (st.stream != stream) HIP_TRY(hipStreamWaitEvent(stream, out ? st.out : st.in, 0));
    return ADSP_OK;
}

// the producer about to fill the slot of step k on `stream` must come after the kernels that read its old contents
int ring_order_producer(adsp_engine* e, hipStream_t stream) {
    if (e->have_last_stream && !e->multi_stream && stream != e->last_stream) {
        int rc = ring_enter_multi_stream(e, stream);
        if (rc) return rc;
    }
    if (!e->multi_stream) return ADSP_OK;
    const long long k = e->step_no, S = e->cfg.ring_slots;
    for (int j = 0; j <= e->cfg.history_chunks; ++j) {
        int rc = ring_wait_step(e, k - S + j, stream, true);
        if (rc) return rc;
    }
    return ADSP_OK;
}
}  // namespace

int adsp_reset(adsp_engine* e) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    ADSP_NOT_LIVE(e);
    int rc = set_device(e);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    e->copy_pending = false;
    ring_forget_steps(e);
    e->lead = 0;
    e->pub_pending = 0;
    HIP_TRY(hipMemset(e->ring, 0, (size_t)e->cfg.ring_slots * e->plane_bytes()));
    e->ring_pos = e->cfg.ring_slots - 1;
    // a fused tremolo starts over as well (the reference pair would be filter.reset + a fresh CreateTremolo)
    e->lfo_copy_len = e->lfo_len;
    e->epi_phase = 0;
    e->epi_replay = 0;
    return ADSP_OK;
}

namespace {
int apply_device_run(adsp_engine* e, const void* d_in, void* d_out, int n_steps, void* stream_v);
}

int adsp_apply_device(adsp_engine* e, const void* d_in, void* d_out, int n_steps, void* stream_v) {
    if (!e || !d_in || !d_out) return fail(ADSP_ERR_ARG, "NULL argument");
    if (n_steps <= 0) return fail(ADSP_ERR_ARG, "n_steps must be positive");
    if (!e->have_spectrum) return fail(ADSP_ERR_STATE, "adsp_set_spectrum has not been called");
    ADSP_NOT_RESIDENT(e);
    int rc = set_device(e);
    if (rc) return rc;
    if (e->epi_op != ADSP_EFFECT_TREMOLO) return apply_device_run(e, d_in, d_out, n_steps, stream_v);
    // the LFO runs on contiguously except where the reference's buffer quirk restarts it: one launch per run
    const size_t plane = e->plane_bytes();
    for (int done = 0; done < n_steps;) {
        const int run = tremolo_run(e, n_steps - done, &e->epi_phase);
        rc = apply_device_run(e, static_cast<const char*>(d_in) + (size_t)done * plane, static_cast<char*>(d_out) + (size_t)done * plane,
                              run, stream_v);
        if (rc) return rc;
        done += run;
    }
    return ADSP_OK;
}

namespace {
int apply_device_run(adsp_engine* e, const void* d_in, void* d_out, int n_steps, void* stream_v) {
    int rc;
    hipStream_t stream = (hipStream_t)stream_v;
    const int S = e->cfg.ring_slots;
    const int cnt = n_steps < e->cfg.history_chunks ? n_steps : e->cfg.history_chunks;
    const size_t plane = e->plane_bytes();
    // The newest `cnt` chunks must end up in the ring.  They go to slots the kernel does not read when the ring has
    // >= 2*history slots, so the copy can run on a side stream BESIDE the kernel: it waits for the caller's input
    // (event on `stream` before the launch) and the next launch on any stream waits for it (event after the copy).
    const bool side = (S >= 2 * e->cfg.history_chunks) && n_steps > 1;
    if (e->multi_stream || (e->have_last_stream && e->last_stream != stream)) {
        // zero-copy steps on other streams may still be in flight: this call joins them, then the step record starts over
        if (!e->ev_join) HIP_TRY(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
        for (auto& st : e->ring_steps)
            if (st.step >= 0 && st.stream != stream) HIP_TRY(hipStreamWaitEvent(stream, st.out, 0));
        if (e->have_last_stream && e->last_stream != stream) {
            HIP_TRY(hipEventRecord(e->ev_join, e->last_stream));
            HIP_TRY(hipStreamWaitEvent(stream, e->ev_join, 0));
        }
        for (auto& st : e->ring_steps) st.step = -1;
        e->multi_stream = false;
    }
    e->have_last_stream = true;
    e->last_stream = stream;
    if (e->copy_pending) {  // a previous side copy must have landed before this kernel reads the ring
        HIP_TRY(hipStreamWaitEvent(stream, e->ev_copy_done, 0));
        e->copy_pending = false;
    }
    if (side) HIP_TRY(hipEventRecord(e->ev_in_ready, stream));
    if ((rc = launch(e, d_in, d_out, n_steps, stream))) return rc;
    hipStream_t cs = side ? e->copy_stream : stream;
    if (side) HIP_TRY(hipStreamWaitEvent(cs, e->ev_in_ready, 0));
    for (int i = 0; i < cnt; ++i) {
        const int slot = (e->ring_pos + 1 + i) % S;
        const char* src = static_cast<const char*>(d_in) + (size_t)(n_steps - cnt + i) * plane;
        HIP_TRY(hipMemcpyAsync(e->ring + (size_t)slot * plane, src, plane, hipMemcpyDefault, cs));  // src: device or mapped host
    }
    if (side) {
        // join: everything the caller enqueues on `stream` after this call (and "stream finished => d_in may be
        // reused") is ordered after the copy, while the copy still overlaps the kernel launched above.
        HIP_TRY(hipEventRecord(e->ev_copy_done, cs));
        HIP_TRY(hipStreamWaitEvent(stream, e->ev_copy_done, 0));
        e->copy_pending = true;  // a later call on a DIFFERENT stream must also wait for it
    }
    e->ring_pos = (e->ring_pos + cnt) % S;
    e->step_no += n_steps;
    e->lead = 0;
    return ADSP_OK;
}
}  // namespace

int adsp_ring_acquire(adsp_engine* e, void** d_slot) {
    if (!e || !d_slot) return fail(ADSP_ERR_ARG, "NULL argument");
    if (e->pipe_depth == 3) return live_pipe_acquire(e, d_slot);
    ADSP_NOT_RESIDENT(e);
    const int slot = (e->ring_pos + 1) % e->cfg.ring_slots;
    *d_slot = e->ring + (size_t)slot * e->plane_bytes();
    if (e->multi_stream) {
        // the caller did not say which stream the producer runs on: the HOST waits for the kernels that still read this slot
        const long long k = e->step_no, S = e->cfg.ring_slots;
        for (int j = 0; j <= e->cfg.history_chunks; ++j) {
            const long long q = k - S + j;
            if (q < 0) continue;
            const auto& st = e->ring_steps[(size_t)(q % (long long)e->ring_steps.size())];
            HIP_TRY(hipEventSynchronize(st.step == q ? st.out : e->ev_join));
        }
    }
    return ADSP_OK;
}

int adsp_ring_reset_order(adsp_engine* e) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    ADSP_NOT_LIVE(e);
    int rc = set_device(e);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    ring_forget_steps(e);
    if (e->lead < 0) e->lead = 0;  // consumer launches that ran ahead have ended (served or timed out)
    return ADSP_OK;
}

int adsp_ring_acquire_stream(adsp_engine* e, void** d_slot, void* stream_v) {
    if (!e || !d_slot) return fail(ADSP_ERR_ARG, "NULL argument");
    if (e->pipe_depth == 3) return live_pipe_acquire(e, d_slot);  // (the session's flow control is the host's: nothing to order on the stream)
    ADSP_NOT_RESIDENT(e);
    int rc = set_device(e);
    if (rc) return rc;
    const int slot = (e->ring_pos + 1) % e->cfg.ring_slots;
    *d_slot = e->ring + (size_t)slot * e->plane_bytes();
    return ring_order_producer(e, (hipStream_t)stream_v);
}

int adsp_apply_ring(adsp_engine* e, void* d_out, void* stream_v) {
    if (!e || !d_out) return fail(ADSP_ERR_ARG, "NULL argument");
    if (!e->have_spectrum) return fail(ADSP_ERR_STATE, "adsp_set_spectrum has not been called");
    if (e->pipe_depth == 3) return live_pipe_apply(e, d_out, (hipStream_t)stream_v);  // the step rides the library's live session
    ADSP_NOT_RESIDENT(e);
    int rc = set_device(e);
    if (rc) return rc;
    if (e->copy_pending) {
        HIP_TRY(hipStreamWaitEvent((hipStream_t)stream_v, e->ev_copy_done, 0));
        e->copy_pending = false;
    }
    const int slot = (e->ring_pos + 1) % e->cfg.ring_slots;
    hipStream_t stream = (hipStream_t)stream_v;
    if (e->pipe_depth > 1) {
        // pipelined: the step runs on the library's stream step % depth, behind an event that marks "everything the caller has
        // enqueued on `stream` so far" - the producer of this step's slot.  The cross-stream ordering of the ring (below) then
        // sees alternating streams exactly as if the caller had alternated them itself.
        hipEvent_t& ev = e->pipe_ev[e->pipe_ev_next++ % 8];
        if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev, stream));
        hipStream_t run = e->pipe_stream[e->step_no % e->pipe_depth];
        HIP_TRY(hipStreamWaitEvent(run, ev, 0));
        stream = run;
    }
    if (e->have_last_stream && !e->multi_stream && stream != e->last_stream && (rc = ring_enter_multi_stream(e, stream))) return rc;
    adsp_engine::RingStep* rec = nullptr;
    if (e->multi_stream) {
        const long long k = e->step_no;
        for (int j = 1; j <= e->cfg.history_chunks; ++j)  // the history this step reads was produced on other streams
            if ((rc = ring_wait_step(e, k - j, stream, false))) return rc;
        rec = &e->ring_steps[(size_t)(k % (long long)e->ring_steps.size())];
        if (!rec->in) {
            HIP_TRY(hipEventCreateWithFlags(&rec->in, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&rec->out, hipEventDisableTiming));
        }
        rec->step = -1;
        HIP_TRY(hipEventRecord(rec->in, stream));  // this step's producer is complete
    }
    if (e->epi_op == ADSP_EFFECT_TREMOLO) (void)tremolo_run(e, 1, &e->epi_phase);
    if ((rc = launch(e, e->ring + (size_t)slot * e->plane_bytes(), d_out, 1, stream))) return rc;
    if (rec) {
        HIP_TRY(hipEventRecord(rec->out, stream));
        rec->step = e->step_no;
        rec->stream = stream;
    }
    e->ring_pos = slot;
    e->step_no += 1;
    if (e->lead > 0) e->lead -= 1;  // a chunk published through adsp_ring_produce_* and consumed step by step
    e->have_last_stream = true;
    e->last_stream = stream;
    return ADSP_OK;
}

int adsp_ring_set_pipeline(adsp_engine* e, int depth) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    if (depth < 1 || depth > 3)
        return fail(ADSP_ERR_ARG, "pipeline depth must be 1 (steps run on the caller's stream), 2 (on the library's two streams in turn) or 3 (ride a live session)");
    ADSP_NOT_RESIDENT(e);  // (also winds down a session the previous depth 3 owned)
    int rc = set_device(e);
    if (rc) return rc;
    if (depth > 1 && e->cfg.ring_slots < e->cfg.history_chunks + 2)
        return fail(ADSP_ERR_ARG, "pipelined steps need ring_slots >= history_chunks + 2 (%d): with fewer the producer of step k + 1 waits for the kernel of step k",
                    e->cfg.history_chunks + 2);
    if (depth == 3 && (rc = live_pipe_check(e))) return rc;  // ADSP_ERR_ARG where no session can run this engine: the caller falls back to depth 2
    HIP_TRY(hipDeviceSynchronize());  // a mode switch: nothing of the ring is in flight
    ring_forget_steps(e);
    for (int i = 0; i < 2 && depth == 2; ++i)
        if (!e->pipe_stream[i]) HIP_TRY(hipStreamCreateWithFlags(&e->pipe_stream[i], hipStreamNonBlocking));
    e->pipe_depth = depth;
    return ADSP_OK;
}

int adsp_ring_join(adsp_engine* e, void* stream_v) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    int rc = set_device(e);
    if (rc) return rc;
    if (e->pipe_depth == 3) {
        // the steps ride a session: the HOST waits until every step submitted so far has its outputs in memory (the publications sit on
        // the caller's stream behind its producers; the session writes through, so any stream may read the outputs afterwards)
        adsp_engine::Live& L = e->live;
        if (!L.active || !L.pipeline_owned || L.published == 0) return ADSP_OK;
        return adsp_live_wait(e, L.published, 20000.0);
    }
    if (e->pipe_depth < 2) return ADSP_OK;  // steps already run on the caller's stream
    for (int i = 0; i < e->pipe_depth; ++i) {
        hipEvent_t& ev = e->pipe_ev[e->pipe_ev_next++ % 8];
        if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev, e->pipe_stream[i]));
        HIP_TRY(hipStreamWaitEvent((hipStream_t)stream_v, ev, 0));
    }
    return ADSP_OK;
}

// ---- resident ring launches -------------------------------------------------------------------------------------
int adsp_ring_produce_begin(adsp_engine* e, void** d_slot, void* stream_v) {
    if (!e || !d_slot) return fail(ADSP_ERR_ARG, "NULL argument");
    ADSP_NOT_LIVE(e);  // (the session owns the ring: resident_prepare would switch its mode, a launch would move ring_pos under adsp_live_slot)
    int rc = set_device(e);
    if (rc) return rc;
    if ((rc = resident_prepare(e))) return rc;
    const int S = e->cfg.ring_slots, h = e->cfg.history_chunks;
    const int ahead = e->lead + e->pub_pending;  // steps handed to the producer and not yet handed to a consumer launch
    if (ahead >= S - h) return fail(ADSP_ERR_STATE, "ring full: %d steps produced and not yet consumed (ring_slots %d - history %d)", ahead, S, h);
    hipStream_t stream = (hipStream_t)stream_v;
    const int slot = (((e->ring_pos + 1 + ahead) % S) + S) % S;
    // the old contents of this slot are step q - S, read by steps q - S .. q - S + h: wait for the resident launches that hold them
    const long long q = e->step_no + ahead;
    for (auto& rl : e->resident_launches)
        if (rl.n > 0 && rl.stream != stream && rl.first <= q - S + h && rl.first + rl.n > q - S && !(rl.waited && rl.waited_by == stream)) {
            HIP_TRY(hipStreamWaitEvent(stream, rl.done, 0));
            rl.waited = true;
            rl.waited_by = stream;
        }
    *d_slot = e->ring + (size_t)slot * e->plane_bytes();
    e->pub_pending += 1;
    return ADSP_OK;
}

int adsp_ring_produce_end(adsp_engine* e, void* stream_v) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    ADSP_NOT_LIVE(e);
    if (!e->resident_mode || e->pub_pending < 1) return fail(ADSP_ERR_STATE, "adsp_ring_produce_end without adsp_ring_produce_begin");
    int rc = set_device(e);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_v;
    const unsigned value = e->pub_count + (unsigned)e->pub_pending;  // every slot handed out since the last publication
    if (!e->seq_by_copy) {
        const hipError_t werr = hipStreamWriteValue32(stream, e->d_seq, value, 0);
        if (werr != hipSuccess) {
            (void)hipGetLastError();
            e->seq_by_copy = true;
            if (getenv("ADSP_DEBUG")) fprintf(stderr, "libadsp: hipStreamWriteValue32 failed (%s): publications become 4-byte copies\n", hipGetErrorString(werr));
        }
    }
    if (e->seq_by_copy) {
        if (!e->pin_seq) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->pin_seq), kSeqPinned * sizeof(unsigned), hipHostMallocDefault));
        unsigned* src = e->pin_seq + value % kSeqPinned;  // reused after kSeqPinned publications: far more than a ring holds
        *src = value;
        HIP_TRY(hipMemcpyAsync(e->d_seq, src, sizeof(unsigned), hipMemcpyHostToDevice, stream));
    }
    // consumer launches that find every one of their steps published run in the tiled workgroup order, in which a workgroup
    // of a later step may be dispatched before one of an earlier step: they must not start before the publications have
    // EXECUTED (not merely been enqueued) - launch() makes the consumer stream wait for this event
    if (!e->ev_pub) HIP_TRY(hipEventCreateWithFlags(&e->ev_pub, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(e->ev_pub, stream));
    e->have_pub = true;
    e->pub_count = value;
    e->lead += e->pub_pending;
    e->pub_pending = 0;
    return ADSP_OK;
}

int adsp_apply_ring_resident(adsp_engine* e, void* d_out, int n_steps, void* stream_v) {
    if (!e || !d_out) return fail(ADSP_ERR_ARG, "NULL argument");
    if (!e->have_spectrum) return fail(ADSP_ERR_STATE, "adsp_set_spectrum has not been called");
    ADSP_NOT_LIVE(e);
    int rc = set_device(e);
    if (rc) return rc;
    if ((rc = resident_prepare(e))) return rc;
    const int S = e->cfg.ring_slots, h = e->cfg.history_chunks;
    if (n_steps < 1 || n_steps > S - h)
        return fail(ADSP_ERR_ARG, "n_steps %d: a resident launch covers 1..ring_slots - history_chunks = %d steps (the slots its own steps do not read)", n_steps, S - h);
    if (e->epi_op == ADSP_EFFECT_TREMOLO) return fail(ADSP_ERR_STATE, "a fused tremolo is not supported by resident launches");
    hipStream_t stream = (hipStream_t)stream_v;
    if (e->copy_pending) {
        HIP_TRY(hipStreamWaitEvent(stream, e->ev_copy_done, 0));
        e->copy_pending = false;
    }
    hipStream_t run = stream;
    if ((rc = launch(e, e->ring, d_out, n_steps, run, true))) return rc;
    // an entry is reusable once its launch has finished (no producer needs to wait for it any more); otherwise the table
    // grows - a large ring consumed by many small launches has many of them in flight, and evicting one would let a
    // producer overwrite a slot that a queued launch has yet to read
    adsp_engine::ResidentLaunch* slot_rl = nullptr;
    for (auto& cand : e->resident_launches)
        if (cand.n == 0 || (cand.done && hipEventQuery(cand.done) == hipSuccess)) {
            slot_rl = &cand;
            break;
        }
    (void)hipGetLastError();
    if (!slot_rl) {
        e->resident_launches.emplace_back();
        slot_rl = &e->resident_launches.back();
    }
    auto& rl = *slot_rl;
    if (!rl.done) HIP_TRY(hipEventCreateWithFlags(&rl.done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(rl.done, run));
    rl.first = e->step_no;
    rl.n = n_steps;
    rl.stream = stream;
    rl.waited = false;
    e->ring_pos = (e->ring_pos + n_steps) % S;
    e->step_no += n_steps;
    e->lead -= n_steps;
    e->have_last_stream = true;
    e->last_stream = stream;
    return ADSP_OK;
}

// ---- live sessions --------------------------------------------------------------------------------------------------
namespace {
// The mapped control words cross PCIe in both directions without any HIP call: plain release stores / acquire loads on the host.
inline void host_word_store(unsigned* p, unsigned v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
inline unsigned host_word_load(const unsigned* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
constexpr int kLiveGpuWords = 128;  // offset (in words) of the GPU-written part of the mapped host words
int live_find_plan(adsp_engine* e, const adsp::LivePlanInfo** out) {
    const adsp_config& c = e->cfg;
    if (e->generic || c.sample_format != ADSP_FORMAT_F32 || c.fft_size != 2 * c.chunk_size)
        return fail(ADSP_ERR_ARG, "live sessions run the stream geometry of float32 engines: power-of-two chunk, fft_size = 2 x chunk_size");
    if (e->epi_op != 0 || e->accumulate != 0) return fail(ADSP_ERR_STATE, "live sessions take no fused effect and no accumulating output");
    const int lq = c.lookback / (c.chunk_size / 4);
    int n = 0;
    const adsp::LivePlanInfo* tab = adsp::live_plans(&n);
    const bool skip8 = getenv("ADSP_LIVE_PLAN16") != nullptr;  // tuning: the 16-points-per-thread plan where the 8-point one would be chosen
    for (int i = 0; i < n; ++i)
        if (tab[i].M == e->M && tab[i].LQ == lq && (c.out_offset / (2 * tab[i].T)) % 2 == 0 && !(skip8 && tab[i].P <= 8)) {  // (the kept rows start on a register pair)
            *out = &tab[i];
            return ADSP_OK;
        }
    return fail(ADSP_ERR_ARG, "no live kernel for chunk %d with lookback %d (= %d quarter chunks): built for chunks 128 .. 4096 with lookback 5/4 N "
                "(cut filters) and 7/4 N (3-band EQ)", c.chunk_size, c.lookback, lq);
}
}  // namespace

int adsp_live_configure(adsp_engine* e, double step_timeout_ms, int load_mode) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    if (e->live.active) return fail(ADSP_ERR_STATE, "a live session is running");
    if (!(step_timeout_ms >= 0.0) || step_timeout_ms > 3.6e6) return fail(ADSP_ERR_ARG, "time-out must be in [0, 3.6e6] ms (0 = wait for ever)");
    if (load_mode < 0 || load_mode > 2) return fail(ADSP_ERR_ARG, "load_mode: 0 plain, 1 non-temporal, 2 system-scope loads");
    e->live.timeout_ms = step_timeout_ms;
    e->live.load_mode = load_mode;
    return ADSP_OK;
}

namespace {
int live_start_impl(adsp_engine* e, void* d_out, int out_slots, unsigned max_steps, void* stream_v, bool with_out_table);
}

int adsp_live_start(adsp_engine* e, void* d_out, int out_slots, unsigned max_steps, void* stream_v) {
    if (!e || !d_out) return fail(ADSP_ERR_ARG, "NULL argument");
    if (!e->have_spectrum) return fail(ADSP_ERR_STATE, "adsp_set_spectrum has not been called");
    if (out_slots < 1 || max_steps < 1) return fail(ADSP_ERR_ARG, "out_slots and max_steps must be positive");
    ADSP_NOT_RESIDENT(e);
    if (e->pipe_depth == 3) return fail(ADSP_ERR_STATE, "ring steps ride a live session of the library's own (adsp_ring_set_pipeline(engine, 3)): switch to depth 1 first");
    return live_start_impl(e, d_out, out_slots, max_steps, stream_v, false);
}

namespace {
int live_start_impl(adsp_engine* e, void* d_out, int out_slots, unsigned max_steps, void* stream_v, bool with_out_table) {
    if (e->multi_stream) return fail(ADSP_ERR_STATE, "ring steps are in flight on several streams: call adsp_ring_reset_order first");
    int rc = set_device(e);
    if (rc) return rc;
    const adsp::LivePlanInfo* lp = nullptr;
    if ((rc = live_find_plan(e, &lp))) return rc;
    adsp_engine::Live& L = e->live;
    const adsp_config& c = e->cfg;
    const int ncg = (c.n_channels + lp->CPB - 1) / lp->CPB;
    // every workgroup of the session must be resident at once: a waiting workgroup that kept another from being dispatched
    // would wait for ever.  The occupancy API may answer one block per CU too many near an SGPR edge (MI355X_MICROARCH.md):
    // one block per CU is left as margin.
    int per_cu = 0, cus = 0;
    HIP_TRY(lp->capacity(&per_cu));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c.device_id));
    const long long room = (long long)(per_cu > 1 ? per_cu - 1 : per_cu) * cus;
    if ((long long)ncg + 2 > room)
        return fail(ADSP_ERR_ARG, "a live session needs all %d workgroups resident at once, this device holds %lld of this kernel (%d per CU, one kept "
                    "as margin): use fewer channels per engine", ncg + 2, room, per_cu);
    if (!L.h_words) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&L.h_words), 2 * kLiveGpuWords * sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent));
        void* d = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&d, L.h_words, 0));
        L.h_words_dev = static_cast<unsigned*>(d);
    }
    if ((size_t)c.n_channels * (size_t)c.chunk_size * sizeof(float) >= 0x7fffffffull)
        return fail(ADSP_ERR_ARG, "a live session addresses a chunk batch with 32-bit byte offsets: channels x chunk must stay below 2 GiB");
    // arrival counters: slot s % A counts the workgroups that have completed step s; A = a power of two beyond the ring, so that
    // no workgroup is ever a whole lap of the counters ahead of the slowest one
    size_t arrival_slots = 1024;
    while (arrival_slots <= (size_t)c.ring_slots) arrival_slots *= 2;
    const size_t n_pad = ((size_t)ncg + 255) & ~(size_t)255;
    const size_t n_words = 4 + n_pad + arrival_slots * 256 + arrival_slots * 2;  // sixteen 64-byte shards per slot; then one 8-byte output address per slot
    if (L.d_words_n < n_words) {
        if (L.d_words) (void)hipFree(L.d_words);
        L.d_words = nullptr;
        L.d_words_n = 0;
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, n_words * sizeof(unsigned), hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            HIP_TRY(hipMalloc(&p, n_words * sizeof(unsigned)));
        }
        L.d_words = static_cast<unsigned*>(p);
        L.d_words_n = n_words;
    }
    hipStream_t stream = (hipStream_t)stream_v;
    if (!stream) {
        // The session's launch never ends while its producer lives, and everything behind it in the same HARDWARE queue waits
        // for it - HIP maps streams onto a handful of hardware queues (measured: every sixth stream created shared the NULL
        // stream's queue, the producer's copy then sat behind the session until the session timed out).  Streams of another
        // priority come from another pool of hardware queues: the session runs on a stream of the highest priority of its own.
        if (!L.own_stream) {
            int least = 0, greatest = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIP_TRY(hipStreamCreateWithPriority(&L.own_stream, hipStreamNonBlocking, greatest));
        }
        stream = L.own_stream;
    }
    if (e->copy_pending) {
        HIP_TRY(hipStreamWaitEvent(stream, e->ev_copy_done, 0));
        e->copy_pending = false;
    }
    // a stream-ordered filter change (adsp_set_spectrum_async) may still be copying the pair tables on the CALLER's stream: the session
    // runs on another one and must not start on half-written tables
    for (int b = 0; b < 2; ++b)
        if (e->tab_busy[b] && e->ev_tab[b]) HIP_TRY(hipStreamWaitEvent(stream, e->ev_tab[b], 0));
    for (int i = 0; i < 2 * kLiveGpuWords; ++i) L.h_words[i] = 0;
    HIP_TRY(hipMemsetAsync(L.d_words, 0, n_words * sizeof(unsigned), stream));
    adsp::LiveArgs la;
    memset(&la, 0, sizeof la);
    adsp::KernelArgs& a = la.k;
    a.ring = e->ring;
    a.tw = e->tw;
    a.pair = e->pair;
    a.pair0 = e->pair0;
    a.zeros = e->zeros;
    a.ring_pos = e->ring_pos;
    a.ring_slots = c.ring_slots;
    a.C = c.n_channels;
    a.n_steps = 1;
    a.V = c.chunk_size;
    a.nblk = 1;
    a.lookback = c.lookback;
    a.j0 = c.out_offset;
    a.ncg = ncg;
    a.N = c.chunk_size;
    a.nh = c.history_chunks;
    a.inv_n = 1.0f / (float)c.chunk_size;
    a.real_spec = e->real_spec ? 1 : 0;
    a.win_pairs = e->plan->P / 2;
    {
        const PlanInfo& ep = *e->plan;
        bool same = ep.P == lp->P && ep.NP == lp->NP && ep.XL == lp->XL && ep.T == lp->T;
        for (int i = 0; i < 4 && same; ++i) same = ep.rad[i] == lp->rad[i];
        if (!same) {
            // the session's plan is not the engine's: its own twiddle and spectrum-stage tables, from the spectrum the engine keeps
            PlanInfo sp = ep;
            sp.P = lp->P, sp.T = lp->T, sp.NP = lp->NP, sp.XL = lp->XL, sp.CPB = lp->CPB, sp.tw_total = lp->tw_total;
            for (int i = 0; i < 4; ++i) sp.rad[i] = lp->rad[i];
            std::vector<float4> tw, tab;
            std::vector<float2> tab0;
            build_twiddles<float>(sp, tw);
            if ((int)tw.size() != lp->tw_total) return fail(ADSP_ERR_STATE, "internal: live plan twiddle count %zu != %d", tw.size(), lp->tw_total);
            if (!e->host_spec.empty()) build_pair_tables<float, float>(sp, e->M, e->host_spec.data(), e->real_spec, tab, tab0);
            else if (!e->host_spec64.empty()) build_pair_tables<float, double>(sp, e->M, e->host_spec64.data(), e->real_spec, tab, tab0);
            else return fail(ADSP_ERR_STATE, "internal: the engine kept no copy of its spectrum");
            auto put = [&](void*& d, size_t& have, const void* src, size_t bytes) -> hipError_t {
                if (have < bytes) {
                    if (d) (void)hipFree(d);
                    d = nullptr;
                    have = 0;
                    hipError_t err = hipMalloc(&d, bytes);
                    if (err != hipSuccess) return err;
                    have = bytes;
                }
                return hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, stream);
            };
            tw.push_back(make_float4(0.f, 0.f, 0.f, 0.f));  // (one entry of padding, like the engine's own table)
            HIP_TRY(put(L.own_tw, L.own_tw_bytes, tw.data(), tw.size() * sizeof(float4)));
            HIP_TRY(put(L.own_pair, L.own_pair_bytes, tab.data(), tab.size() * sizeof(float4)));
            HIP_TRY(put(L.own_pair0, L.own_pair0_bytes, tab0.data(), tab0.size() * sizeof(float2)));
            HIP_TRY(hipStreamSynchronize(stream));  // (the host vectors go out of scope)
            a.tw = L.own_tw;
            a.pair = L.own_pair;
            a.pair0 = L.own_pair0;
        }
    }
    la.out = d_out;
    la.out_slots = out_slots;
    la.first_pub = 0;
    la.max_steps = max_steps;
    la.seq = L.d_words;
    la.done = L.d_words + 1;
    la.stop = L.d_words + 2;
    la.fail = L.d_words + 3;
    la.progress = L.d_words + 4;
    la.arrivals = L.d_words + 4 + n_pad;
    la.arrival_slots = (unsigned)arrival_slots;
    la.host_seq = L.h_words_dev;
    la.host_done = L.h_words_dev + kLiveGpuWords;
    la.host_stop = L.h_words_dev + 2;
    la.timeout = (unsigned long long)(L.timeout_ms * 1e5);  // 100 MHz ticks
    la.load_mode = L.load_mode;
    la.trace = nullptr;
    la.relay_mode = getenv("ADSP_LIVE_RELAY_OFF") ? 1 : 0;
    L.d_out_table = reinterpret_cast<unsigned long long*>(L.d_words + 4 + n_pad + arrival_slots * 256);  // (8-byte aligned: every term is a multiple of 4 words)
    L.out_table_mask = (unsigned)arrival_slots - 1u;
    la.out_table = with_out_table ? L.d_out_table : nullptr;
    la.out_table_mask = L.out_table_mask;
    if (getenv("ADSP_LIVE_TRACE")) {
        if (!L.trace) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&L.trace), 64 * 8 * sizeof(unsigned long long), hipHostMallocMapped));
        memset(L.trace, 0, 64 * 8 * sizeof(unsigned long long));
        void* d = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&d, L.trace, 0));
        la.trace = static_cast<unsigned long long*>(d);
        la.trace_first = (unsigned)atoi(getenv("ADSP_LIVE_TRACE"));
        la.trace_wg = getenv("ADSP_LIVE_TRACE_WG") ? atoi(getenv("ADSP_LIVE_TRACE_WG")) : 0;
        if (la.trace_wg < 0) la.trace_wg += ncg;
    }
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    if (e->timing) {
        if (!e->free_ev.empty()) {
            ev = e->free_ev.back();
            e->free_ev.pop_back();
        } else {
            HIP_TRY(hipEventCreate(&ev.first));
            HIP_TRY(hipEventCreate(&ev.second));
        }
        HIP_TRY(hipEventRecord(ev.first, stream));
    }
    HIP_TRY(lp->launch(la, ncg + 2, stream));  // the workers, then the two relay blocks
    if (e->timing) {
        HIP_TRY(hipEventRecord(ev.second, stream));
        e->timed.push_back(ev);
    }
    L.active = true;
    L.pipeline_owned = false;
    L.plan = lp;
    L.published = L.pending = 0;
    L.max_steps = max_steps;
    L.out_slots = out_slots;
    L.ncg = ncg;
    L.stream = stream;
    return ADSP_OK;
}
}  // namespace

int adsp_live_slot(adsp_engine* e, void** d_slot) {
    if (!e || !d_slot) return fail(ADSP_ERR_ARG, "NULL argument");
    adsp_engine::Live& L = e->live;
    if (!L.active) return fail(ADSP_ERR_STATE, "no live session (adsp_live_start)");
    const unsigned q = L.published + L.pending;  // session index of the step this slot will carry
    if (q >= L.max_steps) return fail(ADSP_ERR_STATE, "the session ends after %u steps", L.max_steps);
    // the slot last carried step q - S (or, for the first lap, a history chunk the kernel loads when it starts): it is free
    // once every workgroup is past step q - (S - history)
    const int S = e->cfg.ring_slots, usable = S - e->cfg.history_chunks;
    const unsigned done = host_word_load(L.h_words + kLiveGpuWords);
    if ((long long)q - usable + 1 > (long long)done)
        return fail(ADSP_ERR_STATE, "ring full: step %u would overwrite a slot the session has not consumed yet (%u steps done, %d usable slots)", q, done, usable);
    const int slot = (int)(((long long)e->ring_pos + 1 + q) % S);
    *d_slot = e->ring + (size_t)slot * e->plane_bytes();
    L.pending += 1;
    return ADSP_OK;
}

int adsp_live_publish_host(adsp_engine* e) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    adsp_engine::Live& L = e->live;
    if (!L.active || L.pending < 1) return fail(ADSP_ERR_STATE, "adsp_live_publish without adsp_live_slot");
    L.published += L.pending;
    L.pending = 0;
    host_word_store(L.h_words, L.published);  // a plain store to mapped memory: no HIP call, no command on any queue
    return ADSP_OK;
}

int adsp_live_publish_stream(adsp_engine* e, void* stream_v) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    adsp_engine::Live& L = e->live;
    if (!L.active || L.pending < 1) return fail(ADSP_ERR_STATE, "adsp_live_publish without adsp_live_slot");
    int rc = set_device(e);
    if (rc) return rc;
    L.published += L.pending;
    L.pending = 0;
    HIP_TRY(adsp::live_publish(L.d_words, L.published, (hipStream_t)stream_v));
    return ADSP_OK;
}

// A data-less producer in a tight native loop (benchmarks, soak tests): the next n_steps slots are taken and published ONE BY
// ONE - whatever the slots hold is the input - waiting for ring space where the session lags.  use_stream: publish through a
// one-lane kernel on `stream` per step; otherwise through host stores.
int adsp_live_publish_run(adsp_engine* e, unsigned n_steps, int use_stream, void* stream_v) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    adsp_engine::Live& L = e->live;
    if (!L.active) return fail(ADSP_ERR_STATE, "no live session (adsp_live_start)");
    const int usable = e->cfg.ring_slots - e->cfg.history_chunks;
    for (unsigned k = 0; k < n_steps; ++k) {
        const unsigned q = L.published + L.pending;
        if (q >= L.max_steps) return fail(ADSP_ERR_STATE, "the session ends after %u steps", L.max_steps);
        if ((long long)q - usable + 1 > (long long)host_word_load(L.h_words + kLiveGpuWords)) {
            const int rc = adsp_live_wait(e, (unsigned)(q - usable + 1), 20000.0);
            if (rc) return rc;
        }
        void* slot = nullptr;
        int rc = adsp_live_slot(e, &slot);
        if (rc) return rc;
        rc = use_stream ? adsp_live_publish_stream(e, stream_v) : adsp_live_publish_host(e);
        if (rc) return rc;
    }
    return ADSP_OK;
}

int adsp_live_progress(adsp_engine* e, unsigned* steps_done) {
    if (!e || !steps_done) return fail(ADSP_ERR_ARG, "NULL argument");
    if (!e->live.h_words) return fail(ADSP_ERR_STATE, "no live session has been started");
    *steps_done = host_word_load(e->live.h_words + kLiveGpuWords);
    return ADSP_OK;
}

int adsp_live_wait(adsp_engine* e, unsigned steps, double timeout_ms) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    adsp_engine::Live& L = e->live;
    if (!L.active) return fail(ADSP_ERR_STATE, "no live session (adsp_live_start)");
    if (steps > L.max_steps) return fail(ADSP_ERR_ARG, "the session ends after %u steps", L.max_steps);
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    unsigned spins = 0;
    const unsigned* h_done = L.h_words + kLiveGpuWords;
    while (host_word_load(h_done) < steps) {
        if ((++spins & 0x3ff) == 0) {
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const double ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
            if (ms > timeout_ms) return fail(ADSP_ERR_STATE, "live session: %u of %u steps done after %.1f ms", host_word_load(h_done), steps, ms);
            if (hipStreamQuery(L.stream) == hipSuccess && host_word_load(h_done) < steps)
                return fail(ADSP_ERR_STATE, "the live session has ended (time-out of a workgroup, or stopped) with %u of %u steps done",
                            host_word_load(h_done), steps);
            (void)hipGetLastError();
        }
    }
    return ADSP_OK;
}

int adsp_live_device_words(adsp_engine* e, unsigned** d_seq, unsigned** d_done) {
    if (!e || !d_seq || !d_done) return fail(ADSP_ERR_ARG, "NULL argument");
    if (!e->live.active) return fail(ADSP_ERR_STATE, "no live session (adsp_live_start)");
    *d_seq = e->live.d_words;
    *d_done = e->live.d_words + 1;
    return ADSP_OK;
}

namespace {
// Ends the session (once every published step is consumed), synchronises its stream and moves the engine's ring on by the steps
// EVERY channel group consumed.  idle_timeout_ok: a session that ended by itself because no step arrived for the configured time-out
// - every workgroup then stands at the last published step - is a clean end, not an error (sessions the pipeline owns).
int live_finish(adsp_engine* e, unsigned* steps_consumed, bool idle_timeout_ok) {
    adsp_engine::Live& L = e->live;
    int rc = set_device(e);
    if (rc) return rc;
    host_word_store(L.h_words + 2, 1u);
    HIP_TRY(hipStreamSynchronize(L.stream));
    std::vector<unsigned> w(4 + (size_t)L.ncg);
    HIP_TRY(hipMemcpy(w.data(), L.d_words, w.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    unsigned done = 0xffffffffu;
    for (int i = 0; i < L.ncg; ++i) done = w[4 + i] < done ? w[4 + i] : done;
    bool timed_out = w[3] != 0;
    if (getenv("ADSP_DEBUG")) {
        fprintf(stderr, "libadsp live_stop: seq %u done %u stop %u fail %u | host_seq %u host_done %u host_stop %u | published %u | progress:", w[0], w[1], w[2],
                w[3], L.h_words[0], L.h_words[kLiveGpuWords], L.h_words[2], L.published);
        const unsigned* g = L.h_words + kLiveGpuWords;
        fprintf(stderr, " h_words %p dev %p d_words %p |", (void*)L.h_words, (void*)L.h_words_dev, (void*)L.d_words);
        fprintf(stderr, " relay: %u iterations, last host_seq %u, exit reason %u |", g[3], g[4], g[6]);
        for (int i = 0; i < L.ncg && i < 64; ++i) fprintf(stderr, " %u", w[4 + i]);
        fprintf(stderr, "\n");
    }
    if (L.trace && getenv("ADSP_LIVE_TRACE")) {
        // average shader cycles between the stamps of workgroup 1 over steps 8 .. 63: top -> chunk requested/waited -> chunk arrived ->
        // window built (+ fetch-ahead issued) -> transform done -> stores issued -> next top
        double seg[6] = {0, 0, 0, 0, 0, 0};
        int n = 0;
        for (int st = 1; st < 63; ++st) {
            const unsigned long long* t = L.trace + st * 8;
            if (!t[0] || !L.trace[(st + 1) * 8]) continue;
            for (int k = 0; k < 5; ++k) seg[k] += (double)(t[k + 1] - t[k]);
            seg[5] += (double)(L.trace[(st + 1) * 8] - t[5]);
            ++n;
        }
        if (n && getenv("ADSP_LIVE_TRACE_RAW"))
            for (int st = 1; st < 25; ++st) {
                const unsigned long long* t = L.trace + st * 8;
                fprintf(stderr, "  step +%d: %llu %llu %llu %llu %llu | next top %llu\n", st, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4],
                        L.trace[(st + 1) * 8] - t[5]);
            }
        if (n)
            fprintf(stderr, "libadsp live trace (workgroup 1, %d steps, shader cycles): wait+request %.0f | chunk arrives %.0f | window %.0f | transform %.0f | "
                    "confirm+stores %.0f | tail %.0f | step %.0f\n", n, seg[0] / n, seg[1] / n, seg[2] / n, seg[3] / n, seg[4] / n, seg[5] / n,
                    (seg[0] + seg[1] + seg[2] + seg[3] + seg[4] + seg[5]) / n);
    }
    L.active = false;
    L.pipeline_owned = false;
    // the ring moves on by the steps EVERY channel group consumed (after a time-out some may be further: adsp_reset then)
    const int S = e->cfg.ring_slots;
    e->ring_pos = (int)(((long long)e->ring_pos + done) % S);
    e->step_no += done;
    e->have_last_stream = true;
    e->last_stream = L.stream;
    if (steps_consumed) *steps_consumed = done;
    if (timed_out && idle_timeout_ok && done == L.published) timed_out = false;  // nothing was pending: every workgroup stands at the same step
    if (timed_out) return fail(ADSP_ERR_STATE, "live session: a workgroup gave up waiting for step %u after %.0f ms (adsp_live_configure); "
                               "the engine's history is undefined: adsp_reset", done, L.timeout_ms);
    return ADSP_OK;
}

// ---- ring steps riding a session (adsp_ring_set_pipeline(engine, 3)) ------------------------------------------------------------
// Winding down a session the pipeline owns: every step the caller has submitted is consumed first (their publications sit on the caller's
// stream and may not have executed yet - stopping at once would drop them), then the session ends.
int live_pipe_release(adsp_engine* e) {
    adsp_engine::Live& L = e->live;
    if (L.published > 0 && host_word_load(L.h_words + kLiveGpuWords + 6) == 0) {  // (still running)
        const int rc = adsp_live_wait(e, L.published, 20000.0);
        if (rc) {
            (void)live_finish(e, nullptr, true);
            return rc;
        }
    }
    return live_finish(e, nullptr, true);
}

// can a session run this engine at all?  (the plan exists and every workgroup is resident at once: what adsp_live_start checks)
int live_pipe_check(adsp_engine* e) {
    const adsp::LivePlanInfo* lp = nullptr;
    int rc = live_find_plan(e, &lp);
    if (rc) return rc;
    const int ncg = (e->cfg.n_channels + lp->CPB - 1) / lp->CPB;
    int per_cu = 0, cus = 0;
    HIP_TRY(lp->capacity(&per_cu));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->cfg.device_id));
    const long long room = (long long)(per_cu > 1 ? per_cu - 1 : per_cu) * cus;
    if ((long long)ncg + 2 > room)
        return fail(ADSP_ERR_ARG, "a live session needs all %d workgroups resident at once, this device holds %lld of this kernel: ring steps of this engine "
                    "cannot ride a session (use pipeline depth 2)", ncg + 2, room);
    return ADSP_OK;
}

// a running session of the pipeline's own: started on first use, restarted when the previous one has ended by itself (idle time-out)
int live_pipe_ensure(adsp_engine* e) {
    adsp_engine::Live& L = e->live;
    if (L.active && !L.pipeline_owned) return fail(ADSP_ERR_STATE, "a live session started with adsp_live_start is running: adsp_live_stop first");
    if (L.active) {
        if (host_word_load(L.h_words + kLiveGpuWords + 6) == 0 && L.published + 1u < L.max_steps) return ADSP_OK;  // (word 6: the relay's exit reason)
        const int rc = live_finish(e, nullptr, true);
        if (rc) return rc;
    }
    if (!e->have_spectrum) return fail(ADSP_ERR_STATE, "adsp_set_spectrum has not been called");
    if (e->resident_mode) return fail(ADSP_ERR_STATE, "the ring is in resident mode: call adsp_ring_reset_order first");
    if (e->multi_stream) return fail(ADSP_ERR_STATE, "ring steps are in flight on several streams: call adsp_ring_reset_order first");
    int rc = set_device(e);
    if (rc) return rc;
    if ((rc = live_start_impl(e, e->ring /* (unused: every step names its own output) */, 1, 0x7fffff00u, nullptr, true))) return rc;
    L.pipeline_owned = true;
    return ADSP_OK;
}

// step q's ring slot may be refilled once the session is past step q - (ring_slots - history): wait for that (the host spins on a mapped word)
int live_pipe_room(adsp_engine* e, unsigned q) {
    adsp_engine::Live& L = e->live;
    const int usable = e->cfg.ring_slots - e->cfg.history_chunks;
    if ((long long)q - usable + 1 > (long long)host_word_load(L.h_words + kLiveGpuWords)) return adsp_live_wait(e, (unsigned)(q - usable + 1), 20000.0);
    return ADSP_OK;
}

int live_pipe_acquire(adsp_engine* e, void** d_slot) {
    int rc = live_pipe_ensure(e);
    if (rc) return rc;
    adsp_engine::Live& L = e->live;
    const unsigned q = L.published;  // the next step (acquiring twice returns the same slot, like the other pipeline depths)
    if ((rc = live_pipe_room(e, q))) return rc;
    *d_slot = e->ring + (size_t)(((long long)e->ring_pos + 1 + q) % e->cfg.ring_slots) * e->plane_bytes();
    return ADSP_OK;
}

int live_pipe_apply(adsp_engine* e, void* d_out, hipStream_t stream) {
    int rc = live_pipe_ensure(e);
    if (rc) return rc;
    adsp_engine::Live& L = e->live;
    const unsigned q = L.published;
    if ((rc = live_pipe_room(e, q))) return rc;  // (a caller that never acquired: the producer is somebody else's business, the ring's is ours)
    // behind whatever filled the slot on `stream`: the step's output address, then the publication
    HIP_TRY(adsp::live_publish_out(L.d_words, q + 1u, L.d_out_table + (q & L.out_table_mask), d_out, stream));
    L.published = q + 1u;
    return ADSP_OK;
}
}  // namespace

int adsp_live_stop(adsp_engine* e, unsigned* steps_consumed) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    adsp_engine::Live& L = e->live;
    if (!L.active) return fail(ADSP_ERR_STATE, "no live session (adsp_live_start)");
    if (L.pipeline_owned) return fail(ADSP_ERR_STATE, "this session belongs to the ring pipeline (adsp_ring_set_pipeline(engine, 3)): switch the depth to end it");
    return live_finish(e, steps_consumed, false);
}


int adsp_ring_resident_timeout(adsp_engine* e, double milliseconds) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    if (!(milliseconds > 0.0) || milliseconds > 60000.0) return fail(ADSP_ERR_ARG, "time-out must be in (0, 60000] ms");
    e->resident_timeout_ticks = (unsigned long long)(milliseconds * 1e5);  // 100 MHz
    return ADSP_OK;
}

int adsp_ring_resident_status(adsp_engine* e, int* timed_out) {
    if (!e || !timed_out) return fail(ADSP_ERR_ARG, "NULL argument");
    *timed_out = 0;
    if (!e->d_seq) return ADSP_OK;
    int rc = set_device(e);
    if (rc) return rc;
    unsigned flag = 0;
    HIP_TRY(hipMemcpy(&flag, e->d_seq + 1, sizeof flag, hipMemcpyDeviceToHost));
    if (flag) {
        const unsigned zero = 0;
        HIP_TRY(hipMemcpy(e->d_seq + 1, &zero, sizeof zero, hipMemcpyHostToDevice));
    }
    *timed_out = flag ? 1 : 0;
    return ADSP_OK;
}

namespace {
constexpr size_t kHostDirectMax = 2u << 20;  // bytes per direction up to which a host call takes the direct path

// One launch, no staging copies: input = pinned host memory the kernel reads over PCIe, output = pinned host memory the
// kernel writes; the call returns as soon as the KERNEL is done (event), the ring update keeps running behind it.
int apply_host_direct(adsp_engine* e, const void* in, void* out, int n_steps, size_t bytes) {
    if (bytes > e->pin_bytes) {
        HIP_TRY(hipDeviceSynchronize());  // nothing may still be reading the old buffers
        for (char** p : {&e->pin_in[0], &e->pin_in[1], &e->pin_out}) {
            if (*p) (void)hipHostFree(*p);
            *p = nullptr;
        }
        e->pin_bytes = 0;
        e->pin_busy[0] = e->pin_busy[1] = false;
        size_t cap = 64u << 10;
        while (cap < bytes) cap *= 2;
        for (char** p : {&e->pin_in[0], &e->pin_in[1], &e->pin_out}) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(p), cap, hipHostMallocMapped));
        for (hipEvent_t* ev : {&e->ev_pin[0], &e->ev_pin[1], &e->ev_kernel})
            if (!*ev) HIP_TRY(hipEventCreate(ev));
        e->pin_bytes = cap;
    }
    const int b = e->pin_slot ^= 1;
    if (e->pin_busy[b]) {  // the ring update two calls ago read this slot
        HIP_TRY(hipEventSynchronize(e->ev_pin[b]));
        e->pin_busy[b] = false;
    }
    memcpy(e->pin_in[b], in, bytes);
    if (e->accumulate) memcpy(e->pin_out, out, bytes);  // the kernel adds to what the output holds
    void *d_in = nullptr, *d_out = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&d_in, e->pin_in[b], 0));
    HIP_TRY(hipHostGetDevicePointer(&d_out, e->pin_out, 0));
    e->want_kernel_event = true;
    const int rc = adsp_apply_device(e, d_in, d_out, n_steps, nullptr);
    e->want_kernel_event = false;
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e->ev_pin[b], nullptr));  // behind the ring update
    e->pin_busy[b] = true;
    HIP_TRY(hipEventSynchronize(e->ev_kernel));
    memcpy(out, e->pin_out, bytes);
    return ADSP_OK;
}
}  // namespace

namespace {
// host memory moved by a few threads at once: one core copies ~10 GB/s, the link takes 63 GB/s each way
void parallel_memcpy(char* dst, const char* src, size_t bytes, int threads) {
    if (threads <= 1 || bytes < (8u << 20)) {
        memcpy(dst, src, bytes);
        return;
    }
    const size_t part = ((bytes / threads) + 4095) & ~(size_t)4095;
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) {
        const size_t off = (size_t)t * part;
        if (off >= bytes) break;
        pool.emplace_back([=] { memcpy(dst + off, src + off, off + part <= bytes ? part : bytes - off); });
    }
    memcpy(dst, src, part < bytes ? part : bytes);
    for (auto& th : pool) th.join();
}


// Hand-over between the three host threads of a pipelined host call (copy in, launch, copy out): counters under one mutex, waiters
// sleep on a condition variable (rounds 4 - 5 spun on atomics with yield(): three cores busy for the length of every large call),
// and the FIRST failure is kept with the hipError_t of the thread it happened on (hipGetLastError is thread-local: the caller's
// would say "no error").
struct PipeSync {
    std::mutex m;
    std::condition_variable cv;
    int staged = 0, issued = 0, drained = 0;
    bool failed = false;
    hipError_t err = hipSuccess;
    const char* where = "";
    void advance(int& counter, int value) {
        {
            std::lock_guard<std::mutex> l(m);
            counter = value;
        }
        cv.notify_all();
    }
    bool wait_for(const int& counter, int at_least) {  // false: somebody failed
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return counter >= at_least || failed; });
        return !failed;
    }
    void fail(hipError_t e, const char* what) {
        {
            std::lock_guard<std::mutex> l(m);
            if (!failed) {
                failed = true;
                err = e;
                where = what;
            }
        }
        cv.notify_all();
    }
    bool has_failed() {
        std::lock_guard<std::mutex> l(m);
        return failed;
    }
};

// The slab pipeline in its default form, without pinned staging of the library's own: a copy-in thread and a copy-out thread give the
// caller's pageable memory to hipMemcpyAsync slab by slab on their own streams; this thread launches the kernels.
int apply_host_direct_slabs(adsp_engine* e, const char* in, char* out, int n_steps, int slab_steps, int n_slabs) {
    adsp_engine::HostPipe& hp = e->hp;
    const size_t step_bytes = e->plane_bytes(), slab_bytes = (size_t)slab_steps * step_bytes;
    const int dev = e->cfg.device_id;
    auto steps_of = [&](int i) { return i + 1 < n_slabs ? slab_steps : n_steps - i * slab_steps; };
    PipeSync ps;
    std::thread stager([&] {
        hipError_t err;
        if ((err = hipSetDevice(dev)) != hipSuccess) return ps.fail(err, "copy-in thread: hipSetDevice");
        for (int i = 0; i < n_slabs; ++i) {
            const int b = i & 1;
            if (i >= 2) {  // d_in[b] was read by the kernel (and the ring update) of slab i - 2
                if (!ps.wait_for(ps.issued, i - 1)) return;
                if ((err = hipEventSynchronize(hp.ev_k[b])) != hipSuccess) return ps.fail(err, "copy-in thread: hipEventSynchronize");
            }
            if ((err = hipMemcpyAsync(hp.d_in[b], in + (size_t)i * slab_bytes, (size_t)steps_of(i) * step_bytes, hipMemcpyHostToDevice, hp.s_in)) != hipSuccess)
                return ps.fail(err, "copy-in thread: hipMemcpyAsync (host to device)");
            if ((err = hipStreamSynchronize(hp.s_in)) != hipSuccess) return ps.fail(err, "copy-in thread: hipStreamSynchronize");
            ps.advance(ps.staged, i + 1);
        }
    });
    std::thread drainer([&] {
        hipError_t err;
        if ((err = hipSetDevice(dev)) != hipSuccess) return ps.fail(err, "copy-out thread: hipSetDevice");
        for (int i = 0; i < n_slabs; ++i) {
            const int b = i & 1;
            if (!ps.wait_for(ps.issued, i + 1)) return;
            if ((err = hipStreamWaitEvent(hp.s_out, hp.ev_k[b], 0)) != hipSuccess) return ps.fail(err, "copy-out thread: hipStreamWaitEvent");
            if ((err = hipMemcpyAsync(out + (size_t)i * slab_bytes, hp.d_out[b], (size_t)steps_of(i) * step_bytes, hipMemcpyDeviceToHost, hp.s_out)) != hipSuccess)
                return ps.fail(err, "copy-out thread: hipMemcpyAsync (device to host)");
            if ((err = hipStreamSynchronize(hp.s_out)) != hipSuccess) return ps.fail(err, "copy-out thread: hipStreamSynchronize");
            ps.advance(ps.drained, i + 1);
        }
    });
    int rc = ADSP_OK;
    for (int i = 0; i < n_slabs; ++i) {
        const int b = i & 1;
        if (!ps.wait_for(ps.staged, i + 1)) break;   // (the copy-in thread synchronised its stream: the data is there)
        if (!ps.wait_for(ps.drained, i - 1)) break;  // d_out[b] has been copied out (slab i - 2)
        if ((rc = adsp_apply_device(e, hp.d_in[b], hp.d_out[b], steps_of(i), hp.s_k))) {
            ps.fail(hipSuccess, "launch thread");
            break;
        }
        const hipError_t herr = hipEventRecord(hp.ev_k[b], hp.s_k);
        if (herr != hipSuccess) {
            ps.fail(herr, "launch thread: hipEventRecord");
            break;
        }
        ps.advance(ps.issued, i + 1);
    }
    stager.join();
    drainer.join();
    (void)hipStreamSynchronize(hp.s_k);
    if (rc) return rc;  // (adsp_apply_device left its own message)
    if (ps.failed) return fail(ADSP_ERR_HIP, "pipelined host call: %s failed: %s", ps.where, hipGetErrorString(ps.err));
    return ADSP_OK;
}

constexpr size_t kPipeSlabTarget = 48u << 20;  // bytes per slab and direction: four pinned + four device buffers of this size per engine

// Large host batches (the numpy API on a real batch: WavBank.process, apply_batch - EffectFFTFilter.py:49-75 for C channels and many
// chunks at once): slabs of whole steps, double-buffered on the device.  Three threads of control on the host - one that copies slabs
// in, this thread that launches, one that copies slabs out - and three streams on the device, so that the H2D copy of slab i + 1, the
// kernel of slab i and the D2H copy of slab i - 1 overlap.  Steps are independent through the engine's history ring, so a slab is just
// a shorter call.  (ADSP_HOST_STAGING=pinned: the same through pinned staging buffers of the library's own, filled and emptied by a few
// host threads - kept for A/B, slower on the boxes measured.)
int apply_host_pipelined(adsp_engine* e, const char* in, char* out, int n_steps) {
    adsp_engine::HostPipe& hp = e->hp;
    const size_t step_bytes = e->plane_bytes();
    int slab_steps = (int)(kPipeSlabTarget / step_bytes);
    if (slab_steps < 1) slab_steps = 1;
    if (slab_steps > (n_steps + 3) / 4) slab_steps = (n_steps + 3) / 4;  // at least four slabs
    if (!e->generic && e->block_outputs > e->cfg.chunk_size) {
        // multi-step launches tile the time axis with block_outputs kept samples: whole tiles per slab (a slab's last block is then full)
        long long tile = e->block_outputs, g = e->cfg.chunk_size;
        for (long long a = tile, b = g; b;) { const long long t = a % b; a = b; b = t; g = a; }
        const int tile_steps = (int)(tile / g);  // lcm(block_outputs, N) / N
        if (slab_steps >= tile_steps) slab_steps = slab_steps / tile_steps * tile_steps;
    }
    const size_t slab_bytes = (size_t)slab_steps * step_bytes;
    // default: no pinned staging of the library's own - a copy-in thread and a copy-out thread hand the caller's pageable memory to
    // hipMemcpyAsync slab by slab (the runtime stages it itself) on two copy streams, overlapped with the kernels and with each other
    // Measured on MI355X (profiles/r5_host_staging.txt, 1 GiB each way): this form 23.8 ms = 45 GB/s per direction (72 % of the link);
    // the library's own pinned staging (ADSP_HOST_STAGING=pinned: pageable -> pinned copies by 2 / 4 / 8 host threads per direction,
    // hipMemcpyAsync from pinned memory) 34.5 / 42.5 / 41.5 ms - the host's memory system, not the link, is what the extra copy costs;
    // the one-piece form of rounds 1 - 4 (pageable hipMemcpy in, kernel, hipMemcpy out) 38.8 ms.
    const char* mode = getenv("ADSP_HOST_STAGING");
    const bool direct = !(mode && strcmp(mode, "pinned") == 0);
    if (hp.slab_bytes < slab_bytes || (!direct && !hp.pin_in[0])) {
        HIP_TRY(hipDeviceSynchronize());
        for (int b = 0; b < 2; ++b) {
            for (char** p : {&hp.pin_in[b], &hp.pin_out[b]}) {
                if (*p) (void)hipHostFree(*p);
                *p = nullptr;
            }
            for (char** p : {&hp.d_in[b], &hp.d_out[b]}) {
                if (*p) (void)hipFree(*p);
                *p = nullptr;
            }
        }
        hp.slab_bytes = 0;
        for (int b = 0; b < 2; ++b) {
            if (!direct) {
                HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&hp.pin_in[b]), slab_bytes, hipHostMallocDefault));
                HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&hp.pin_out[b]), slab_bytes, hipHostMallocDefault));
            }
            HIP_TRY(hipMalloc(&hp.d_in[b], slab_bytes));
            HIP_TRY(hipMalloc(&hp.d_out[b], slab_bytes));
        }
        hp.slab_bytes = slab_bytes;
    }
    if (!hp.s_in) {
        HIP_TRY(hipStreamCreateWithFlags(&hp.s_in, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&hp.s_k, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&hp.s_out, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            HIP_TRY(hipEventCreateWithFlags(&hp.ev_in[b], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&hp.ev_k[b], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&hp.ev_out[b], hipEventDisableTiming));
        }
    }
    HIP_TRY(hipStreamSynchronize(nullptr));  // earlier calls of this engine on the default stream (the small-call path) are complete
    const int n_slabs = (n_steps + slab_steps - 1) / slab_steps;
    if (direct) return apply_host_direct_slabs(e, in, out, n_steps, slab_steps, n_slabs);
    const int dev = e->cfg.device_id;
    unsigned hw = std::thread::hardware_concurrency();
    int copy_threads = hw >= 16 ? 4 : hw >= 8 ? 2 : 1;
    if (const char* t = getenv("ADSP_HOST_COPY_THREADS")) copy_threads = atoi(t) > 0 && atoi(t) <= 32 ? atoi(t) : copy_threads;  // (tuning)
    auto steps_of = [&](int i) { return i + 1 < n_slabs ? slab_steps : n_steps - i * slab_steps; };
    PipeSync ps;
    // stager: slab i -> pin_in[i % 2] once the H2D copy of slab i - 2 has left it
    std::thread stager([&] {
        hipError_t err;
        if ((err = hipSetDevice(dev)) != hipSuccess) return ps.fail(err, "staging thread: hipSetDevice");
        for (int i = 0; i < n_slabs; ++i) {
            const int b = i & 1;
            if (i >= 2) {
                if (!ps.wait_for(ps.issued, i - 1)) return;  // (its copy has been enqueued: the event is recorded)
                if ((err = hipEventSynchronize(hp.ev_in[b])) != hipSuccess) return ps.fail(err, "staging thread: hipEventSynchronize");
            }
            parallel_memcpy(hp.pin_in[b], in + (size_t)i * slab_bytes, (size_t)steps_of(i) * step_bytes, copy_threads);
            ps.advance(ps.staged, i + 1);
        }
    });
    // drainer: pin_out[i % 2] -> the caller's array once the D2H copy of slab i has landed
    std::thread drainer([&] {
        hipError_t err;
        if ((err = hipSetDevice(dev)) != hipSuccess) return ps.fail(err, "draining thread: hipSetDevice");
        for (int i = 0; i < n_slabs; ++i) {
            const int b = i & 1;
            if (!ps.wait_for(ps.issued, i + 1)) return;
            if ((err = hipEventSynchronize(hp.ev_out[b])) != hipSuccess) return ps.fail(err, "draining thread: hipEventSynchronize");
            parallel_memcpy(out + (size_t)i * slab_bytes, hp.pin_out[b], (size_t)steps_of(i) * step_bytes, copy_threads);
            ps.advance(ps.drained, i + 1);
        }
    });
    int rc = ADSP_OK;
    hipError_t herr = hipSuccess;
    const char* at = "";
    for (int i = 0; i < n_slabs && rc == ADSP_OK && herr == hipSuccess; ++i) {
        const int b = i & 1, ns = steps_of(i);
        const size_t bytes = (size_t)ns * step_bytes;
        if (!ps.wait_for(ps.staged, i + 1)) break;
        // d_in[b] was read by the kernel (and the ring update) of slab i - 2; pin_out[b] / d_out[b] must have been drained of slab i - 2
        at = "launch thread: copy in";
        if (i >= 2 && (herr = hipStreamWaitEvent(hp.s_in, hp.ev_k[b], 0)) != hipSuccess) break;
        if ((herr = hipMemcpyAsync(hp.d_in[b], hp.pin_in[b], bytes, hipMemcpyHostToDevice, hp.s_in)) != hipSuccess) break;
        if ((herr = hipEventRecord(hp.ev_in[b], hp.s_in)) != hipSuccess) break;
        if ((herr = hipStreamWaitEvent(hp.s_k, hp.ev_in[b], 0)) != hipSuccess) break;
        if (i >= 2 && (herr = hipStreamWaitEvent(hp.s_k, hp.ev_out[b], 0)) != hipSuccess) break;  // d_out[b]: the D2H copy of slab i - 2 is done
        if ((rc = adsp_apply_device(e, hp.d_in[b], hp.d_out[b], ns, hp.s_k))) break;
        at = "launch thread: copy out";
        if ((herr = hipEventRecord(hp.ev_k[b], hp.s_k)) != hipSuccess) break;
        if (!ps.wait_for(ps.drained, i - 1)) break;  // pin_out[b] has been copied out (slab i - 2)
        if ((herr = hipStreamWaitEvent(hp.s_out, hp.ev_k[b], 0)) != hipSuccess) break;
        if ((herr = hipMemcpyAsync(hp.pin_out[b], hp.d_out[b], bytes, hipMemcpyDeviceToHost, hp.s_out)) != hipSuccess) break;
        if ((herr = hipEventRecord(hp.ev_out[b], hp.s_out)) != hipSuccess) break;
        ps.advance(ps.issued, i + 1);
    }
    if (rc != ADSP_OK || herr != hipSuccess) ps.fail(herr, at);
    stager.join();
    drainer.join();
    (void)hipStreamSynchronize(hp.s_in);
    (void)hipStreamSynchronize(hp.s_k);
    (void)hipStreamSynchronize(hp.s_out);
    if (rc) return rc;
    if (ps.failed) return fail(ADSP_ERR_HIP, "pipelined host call: %s failed: %s", ps.where, hipGetErrorString(ps.err));
    return ADSP_OK;
}
}  // namespace

int adsp_apply_host(adsp_engine* e, const void* in, void* out, int n_steps) {
    if (!e || !in || !out) return fail(ADSP_ERR_ARG, "NULL argument");
    if (n_steps <= 0) return fail(ADSP_ERR_ARG, "n_steps must be positive");
    if (!e->have_spectrum) return fail(ADSP_ERR_STATE, "adsp_set_spectrum has not been called");
    ADSP_NOT_RESIDENT(e);
    int rc = set_device(e);
    if (rc) return rc;
    const size_t elems = (size_t)n_steps * e->plane();
    if (elems * e->ssize() <= kHostDirectMax) return apply_host_direct(e, in, out, n_steps, elems * e->ssize());
    // real batches: slabs through pinned staging, copies and kernels overlapped (a fused tremolo and an accumulating output keep the
    // one-piece form: the first restarts its LFO per launch run, the second needs the caller's output on the device first)
    if (n_steps >= 4 && elems * e->ssize() >= (16u << 20) && e->accumulate == 0 && e->epi_op != ADSP_EFFECT_TREMOLO && !getenv("ADSP_HOST_UNPIPELINED"))
        return apply_host_pipelined(e, static_cast<const char*>(in), static_cast<char*>(out), n_steps);
    if (elems > e->stage_elems) {
        HIP_TRY(hipDeviceSynchronize());
        if (e->stage_in) (void)hipFree(e->stage_in);
        if (e->stage_out) (void)hipFree(e->stage_out);
        e->stage_in = e->stage_out = nullptr;
        e->stage_elems = 0;
        HIP_TRY(hipMalloc(&e->stage_in, elems * e->ssize()));
        HIP_TRY(hipMalloc(&e->stage_out, elems * e->ssize()));
        e->stage_elems = elems;
    }
    HIP_TRY(hipMemcpy(e->stage_in, in, elems * e->ssize(), hipMemcpyHostToDevice));
    if (e->accumulate) HIP_TRY(hipMemcpy(e->stage_out, out, elems * e->ssize(), hipMemcpyHostToDevice));
    if ((rc = adsp_apply_device(e, e->stage_in, e->stage_out, n_steps, nullptr))) return rc;
    HIP_TRY(hipMemcpy(out, e->stage_out, elems * e->ssize(), hipMemcpyDeviceToHost));
    return ADSP_OK;
}

int adsp_get_state(adsp_engine* e, void* host_history) {
    if (!e || !host_history) return fail(ADSP_ERR_ARG, "NULL argument");
    ADSP_NOT_LIVE(e);
    int rc = set_device(e);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    e->copy_pending = false;
    ring_forget_steps(e);
    const int S = e->cfg.ring_slots, nh = e->cfg.history_chunks;
    const size_t plane = e->plane_bytes();
    for (int h = 0; h < nh; ++h) {  // h = 0 oldest (time step -nh)
        const int slot = ((e->ring_pos + 1 - nh + h) % S + S) % S;
        HIP_TRY(hipMemcpy(static_cast<char*>(host_history) + (size_t)h * plane, e->ring + (size_t)slot * plane, plane, hipMemcpyDeviceToHost));
    }
    return ADSP_OK;
}

int adsp_set_state(adsp_engine* e, const void* host_history) {
    if (!e || !host_history) return fail(ADSP_ERR_ARG, "NULL argument");
    ADSP_NOT_LIVE(e);
    int rc = set_device(e);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    const int S = e->cfg.ring_slots, nh = e->cfg.history_chunks;
    const size_t plane = e->plane_bytes();
    for (int h = 0; h < nh; ++h) {
        const int slot = ((e->ring_pos + 1 - nh + h) % S + S) % S;
        HIP_TRY(hipMemcpy(e->ring + (size_t)slot * plane, static_cast<const char*>(host_history) + (size_t)h * plane, plane, hipMemcpyHostToDevice));
    }
    return ADSP_OK;
}

int adsp_enable_kernel_timing(adsp_engine* e, int enable) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    e->timing = enable != 0;
    return ADSP_OK;
}

int adsp_kernel_time(adsp_engine* e, double* total_ms, int* launches) {
    if (!e || !total_ms || !launches) return fail(ADSP_ERR_ARG, "NULL argument");
    int rc = set_device(e);
    if (rc) return rc;
    double sum = 0.0;
    for (auto& p : e->timed) {
        HIP_TRY(hipEventSynchronize(p.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
        sum += ms;
    }
    *total_ms = sum;
    *launches = (int)e->timed.size();
    e->free_ev.insert(e->free_ev.end(), e->timed.begin(), e->timed.end());
    e->timed.clear();
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

int adsp_synchronize(adsp_engine* e, void* stream) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    int rc = set_device(e);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return ADSP_OK;
}

}  // extern "C"
