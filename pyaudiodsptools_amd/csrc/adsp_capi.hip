// adsp_capi.hip - C ABI (include/adsp.h) over the fused overlap-save kernel.
// Host logic only: plan registry, twiddle / spectrum-pair tables (float64 -> float32), the input
// history ring, launches.  There is deliberately NO CPU fallback: without a GPU every compute
// entry point fails with ADSP_ERR_NO_DEVICE / ADSP_ERR_HIP.
#include "engine_internal.hpp"

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
using namespace adsp::tables;
namespace {

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

namespace adsp_internal {

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

int launch(adsp_engine* e, const void* d_in, void* d_out, int n_steps, hipStream_t stream, bool resident) {
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
    if (pl.M >= 16384 && !e->generic && n_steps > 1 && !(getenv("ADSP_NT_HYBRID") && atoi(getenv("ADSP_NT_HYBRID")) == 0)) {  // (ADSP_NT_HYBRID=0: tuning A/B)
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

}  // namespace adsp_internal

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

}  // extern "C"
namespace adsp_internal {
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
}  // namespace adsp_internal
extern "C" {

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


}  // extern "C"
namespace adsp_internal {
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
}  // namespace adsp_internal
extern "C" {

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

int adsp_synchronize(adsp_engine* e, void* stream) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    int rc = set_device(e);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return ADSP_OK;
}

}  // extern "C"
