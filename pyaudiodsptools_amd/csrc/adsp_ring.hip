// adsp_ring.hip - the real-time call patterns of include/adsp.h on one engine: zero-copy ring steps (stream ordering across streams,
// library-pipelined steps), resident launches (one launch consumes many published steps), live sessions (ONE persistent launch) and
// ring steps riding a session (adsp_ring_set_pipeline(engine, 3)).  The engine object and the launch helper: engine_internal.hpp.
#include "engine_internal.hpp"

using namespace adsp::tables;

namespace adsp_internal {
int live_start_impl(adsp_engine* e, void* d_out, int out_slots, unsigned max_steps, void* stream_v, bool with_out_table);
}  // namespace adsp_internal

namespace adsp_internal {
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
}  // namespace adsp_internal

extern "C" {

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
}  // extern "C"
namespace adsp_internal {
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
}  // namespace adsp_internal
extern "C" {

int adsp_live_configure(adsp_engine* e, double step_timeout_ms, int load_mode) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    if (e->live.active) return fail(ADSP_ERR_STATE, "a live session is running");
    if (!(step_timeout_ms >= 0.0) || step_timeout_ms > 3.6e6) return fail(ADSP_ERR_ARG, "time-out must be in [0, 3.6e6] ms (0 = wait for ever)");
    if (load_mode < 0 || load_mode > 2) return fail(ADSP_ERR_ARG, "load_mode: 0 plain, 1 non-temporal, 2 system-scope loads");
    e->live.timeout_ms = step_timeout_ms;
    e->live.load_mode = load_mode;
    return ADSP_OK;
}


int adsp_live_start(adsp_engine* e, void* d_out, int out_slots, unsigned max_steps, void* stream_v) {
    if (!e || !d_out) return fail(ADSP_ERR_ARG, "NULL argument");
    if (!e->have_spectrum) return fail(ADSP_ERR_STATE, "adsp_set_spectrum has not been called");
    if (out_slots < 1 || max_steps < 1) return fail(ADSP_ERR_ARG, "out_slots and max_steps must be positive");
    ADSP_NOT_RESIDENT(e);
    if (e->pipe_depth == 3) return fail(ADSP_ERR_STATE, "ring steps ride a live session of the library's own (adsp_ring_set_pipeline(engine, 3)): switch to depth 1 first");
    return live_start_impl(e, d_out, out_slots, max_steps, stream_v, false);
}

}  // extern "C"
namespace adsp_internal {
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
}  // namespace adsp_internal
extern "C" {

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

}  // extern "C"
namespace adsp_internal {
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
}  // namespace adsp_internal
extern "C" {

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
}  // extern "C"
