// engine_internal.hpp - what the translation units behind include/adsp.h's engine entry points share: the engine object and the
// helpers that cross file boundaries.  adsp_capi.hip: plans, tables, engine lifecycle, spectrum, launches, state;  adsp_ring.hip: the
// zero-copy ring (stream ordering, pipelined steps), resident launches, live sessions;  adsp_host.hip: host batches (direct, staged,
// pipelined slabs);  adsp_effects.hip: the standalone elementwise kernels.  Round 6 split of what was one 2500-line file.
#pragma once
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

using adsp::fail;
using adsp::PlanInfo;

namespace adsp {  // adsp_rccl.hip
int rccl_broadcast(float* const* d_buf, const int* devs, const hipStream_t* streams, int n, size_t count, int root);
int rccl_version(int* version);
int rccl_unique_id(char* out);
int rccl_broadcast_rank(const char* unique_id, int rank, int world, int root, int dev, float* d_buf, size_t count, hipStream_t stream);
int rccl_finalize();
}  // namespace adsp

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

namespace adsp_internal {
// adsp_capi.hip
int set_device(const adsp_engine* e);
int launch(adsp_engine* e, const void* d_in, void* d_out, int n_steps, hipStream_t stream, bool resident = false);
int prepare_twin(adsp_engine* e);
int tremolo_run(adsp_engine* e, int max_steps, int* phase);
int apply_device_run(adsp_engine* e, const void* d_in, void* d_out, int n_steps, void* stream_v);
// adsp_ring.hip
void ring_forget_steps(adsp_engine* e);
int resident_prepare(adsp_engine* e);
int ring_enter_multi_stream(adsp_engine* e, hipStream_t stream);
int ring_wait_step(adsp_engine* e, long long k, hipStream_t stream, bool out);
int ring_order_producer(adsp_engine* e, hipStream_t stream);
// A session started by the library itself (adsp_ring_set_pipeline(engine, 3): adsp_apply_ring rides a live session) is wound down by
// any call that needs the engine in its ordinary state; a session the caller started (adsp_live_start) is the caller's to stop.
int live_pipe_release(adsp_engine* e);
int live_pipe_acquire(adsp_engine* e, void** d_slot);
int live_pipe_apply(adsp_engine* e, void* d_out, hipStream_t stream);
int live_pipe_check(adsp_engine* e);
}  // namespace adsp_internal
using namespace adsp_internal;

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
