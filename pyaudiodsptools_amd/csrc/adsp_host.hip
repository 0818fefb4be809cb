// adsp_host.hip - host batches (include/adsp.h: adsp_apply_host): small calls through pinned, device-mapped memory without staging
// copies; larger ones through device staging; real batches in slabs, the copy-in, launch and copy-out of consecutive slabs overlapped
// by three host threads on three streams (handing over through a mutex and a condition variable).
#include "engine_internal.hpp"

namespace adsp_internal {
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
}  // namespace adsp_internal

namespace adsp_internal {
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
}  // namespace adsp_internal

extern "C" {

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

}  // extern "C"
