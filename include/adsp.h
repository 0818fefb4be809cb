/*
 * adsp.h - C ABI of the MI355X-native batched FFT filter / FFT-EQ engine (libadsp.so).
 *
 * This is the drop-in boundary of the hot path.  The reference (ArjaanAuinger/pyaudiodsptools) is
 * pure Python and has no FFI of its own; the entry points below are what a Python binding for
 *
 *     CreateHighCutFilter.apply / CreateLowCutFilter.apply   pyAudioDspTools/EffectFFTFilter.py:49-75, :125-151
 *     CreateEQ3BandFFT.apply                                  pyAudioDspTools/EffectEQ3BandFFT.py:156-211
 *
 * binds (see INTEGRATION.md for the ctypes stub).  The reference's apply() is, exactly, a
 * streaming FIR with one chunk of latency; this library computes the same
 * stream for C independent mono channels at once with an overlap-save real FFT executed by
 * hand-written HIP kernels for gfx950.  The engine is filter-agnostic: the host designs the FIR
 * (float64), hands over its spectrum, and says where in the transform window the kept samples
 * live.
 *
 * Conventions
 *   - every function returns 0 on success, a negative adsp_status on failure; the message is
 *     available from adsp_last_error() (thread-local).  No exception crosses this ABI.
 *   - plain pointers and sizes only; no torch / numpy types.
 *   - one host thread per engine at a time (the reference's devices are not thread-safe either,
 *     EffectFFTFilter.py:63-65); distinct engines are independent.
 *   - the library owns all device memory it allocates; "d_" pointers are caller-owned device
 *     memory, all others are host memory.
 *   - batch layout everywhere: [step][channel][sample] row-major, samples in the engine's
 *     sample_format (float32 or int16), i.e. the chunk of channel c at step k starts at sample
 *     ((k * n_channels) + c) * chunk_size.  Device buffers must be 16-byte aligned.
 */
#ifndef ADSP_H
#define ADSP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define ADSP_API __attribute__((visibility("default")))
#else
#define ADSP_API
#endif

#define ADSP_ABI_VERSION 11
#define ADSP_MAX_HISTORY 8

typedef enum adsp_status {
    ADSP_OK = 0,
    ADSP_ERR_ARG = -1,        /* bad argument / unsupported geometry */
    ADSP_ERR_HIP = -2,        /* a HIP runtime call failed */
    ADSP_ERR_STATE = -3,      /* call sequence error (e.g. apply before set_spectrum) */
    ADSP_ERR_NO_DEVICE = -4   /* no usable GPU */
} adsp_status;

typedef struct adsp_engine adsp_engine; /* opaque.  Engines (and delay lines, scans) are not internally locked: use one from
                                           one thread at a time; different engines are independent. */

/* Sample formats of the [step][channel][sample] batches an engine filters (adsp_config.sample_format). */
#define ADSP_FORMAT_F32 0 /* float32, the reference's in-memory format */
#define ADSP_FORMAT_S16 1 /* int16 PCM, the reference's WAV format: the kernel converts (float)x on load and
                             (int16)trunc(y) on store - Utility.py:233-238 (/32768) and :295-312 (*32767,
                             astype int16) become ONE factor 32767/32768 folded into the spectrum by the host */
#define ADSP_FORMAT_S16_F64 2 /* int16 PCM batches filtered in FLOAT64 (the float64 instantiation of the same kernels): the
                                 "exact FFT" engines.  The conversions are the reference's to the letter - float32(pcm) / 32768
                                 in, int16(trunc(float32(y) * 32767)) out (Utility.py:236-237, :306; the spectrum carries NO
                                 gain) - and the float64 transform's error (1e-15 of full scale) is far below the reference's
                                 own, so the int16 stream equals the float64 direct sum of adsp_exact_* except where y lies
                                 within ~1e-15 of a float32 rounding boundary (about one sample in 1e7).  Give the spectrum
                                 with adsp_set_spectrum_f64.  About a third of the float32 engines' rate. */

/*
 * Geometry of one streaming FIR engine.
 *
 * For an output block that starts at output-time o (o = 0 is the first sample returned by the
 * first apply() after reset), the kernel transforms the fft_size input samples starting at
 * input-time o - lookback (input-time 0 is the first sample ever fed; earlier samples are the
 * zero history of EffectFFTFilter.py:40-42), multiplies by the spectrum and keeps block_outputs
 * samples starting at circular index out_offset of the inverse transform.
 *
 * Single reference device with chunk N (L = N/2-1 taps, d = (L-1)/2):
 *   low/high cut : fft_size 2N, history_chunks 2, lookback N + N/4, out_offset N/4  (symmetric kernel centred on
 *                  circular index 0: its spectrum is real, which adsp_set_spectrum detects - every imaginary part
 *                  exactly 0 - and answers with a cheaper spectrum stage, 3 real constants per bin pair)
 *   3-band EQ    : fft_size 2N, history_chunks 2, lookback 2N - N/4, out_offset N  (kernel delayed by 1 tap)
 */
typedef struct adsp_config {
    int device_id;       /* HIP device ordinal */
    int chunk_size;      /* N: samples per channel per step; any length >= 4 (the reference takes any chunk_size: EffectFFTFilter.py:22).
                            Multiples of 4 from 16 up move 16 bytes per access; other lengths run the generic kernel's dword-access
                            form (float32 engines only, no fused effect) */
    int n_channels;      /* C: independent mono channels held by this engine */
    int fft_size;        /* F: real transform length, power of two in 128..32768.  N a power of two in 64..8192 with
                            F = 2N or 4N selects the specialised kernels (chunk boundaries known at compile time);
                            anything else runs the generic-geometry kernel */
    int history_chunks;  /* past chunks the window can reach (1..ADSP_MAX_HISTORY) */
    int lookback;        /* see above; 0 < lookback <= history_chunks*N; multiple of N/4 (specialised) or 4 (generic; any value
                            when the chunk size is not a multiple of 4) */
    int out_offset;      /* see above; multiple of N/4 (specialised) or 4*threads_per_transform (generic) */
    int ring_slots;      /* input ring length (>= history_chunks+1); 0 = 2*history_chunks, which lets the ring
                            update of multi-step launches run on a side stream beside the kernel */
    int sample_format;   /* ADSP_FORMAT_F32, ADSP_FORMAT_S16 or ADSP_FORMAT_S16_F64 (int16 buffers): type of every `in`/`out`/ring/state buffer below */
} adsp_config;

/* ABI version (ADSP_ABI_VERSION of the built library). */
ADSP_API int adsp_version(void);

/* "product": libadsp.so, what `make` builds and the package loads.  "tuning": libadsp_tuning.so (`make tuning`, loaded only through
 * ADSP_LIB=<path>): the same entry points plus the A/B plan variants (ADSP_PLAN_VARIANT), the ablation switches and the persistent-block
 * kernels of the speed-of-light model - none of which exist in the product library, which refuses ADSP_PLAN_VARIANT with a message. */
ADSP_API const char* adsp_build_info(void);

/* Thread-local description of the last failure on this thread ("" if none). */
ADSP_API const char* adsp_last_error(void);

/* Number of visible GPUs (0 and ADSP_ERR_NO_DEVICE when there is none). */
ADSP_API int adsp_device_count(int* count);

/* 0 if a kernel exists for this (chunk_size, fft_size), ADSP_ERR_ARG otherwise.  Needs no GPU. */
ADSP_API int adsp_plan_supported(int chunk_size, int fft_size);

/* Describe the plan chosen for (chunk_size, fft_size): complex points M, points per thread,
 * threads per transform, channels per workgroup, LDS bytes per workgroup.  Needs no GPU. */
ADSP_API int adsp_plan_describe(int chunk_size, int fft_size, int* complex_points, int* points_per_thread,
                       int* threads_per_transform, int* channels_per_workgroup, int* lds_bytes);

/* Allocate an engine: zeroed input ring [ring_slots][C][N], twiddle tables.  Replaces the
 * reference constructors' state setup (EffectFFTFilter.py:39-42; EffectEQ3BandFFT.py:147-152). */
ADSP_API int adsp_create(const adsp_config* cfg, adsp_engine** out_engine);
ADSP_API int adsp_destroy(adsp_engine* engine);

/* Upload the filter spectrum: n_bins = F/2+1 interleaved (re,im) float32 values of rfft(kernel, F),
 * computed by the host in float64.  Replaces self.sinc_filter (EffectFFTFilter.py:45-47) and the
 * four EQ spectra + per-call recombination (EffectEQ3BandFFT.py:88-143, :182-188).  May be called
 * again at any time to change the filter (takes effect for the next apply). */
ADSP_API int adsp_set_spectrum(adsp_engine* engine, const float* spectrum_interleaved, int n_bins);

/* The same from a float64 spectrum (interleaved re, im doubles): what ADSP_FORMAT_S16_F64 engines want - their tables are
 * built and kept in float64; a float32 / int16 engine rounds it to float32 first.  Set-up path like adsp_set_spectrum. */
ADSP_API int adsp_set_spectrum_f64(adsp_engine* engine, const double* spectrum_interleaved, int n_bins);

/* The same filter change in a LIVE stream: the tables are staged in pinned memory and copied on `stream`, so launches
 * already queued there finish with the old filter, later ones use the new one, and nothing else is waited for (no
 * device-wide synchronisation; adsp_set_spectrum drains the device and is meant for set-up).  Launches that follow on
 * OTHER streams must be ordered after `stream` by the caller. */
ADSP_API int adsp_set_spectrum_async(adsp_engine* engine, const float* spectrum_interleaved, int n_bins, void* stream);

/* Same, from device memory (e.g. after an RCCL broadcast enqueued on `stream`): waits for `stream` only - the tables are
 * float64 host arithmetic, so the spectrum makes one trip to the host - then updates them stream-ordered as above.
 * Both stream-ordered updates rewrite the tables in place: NO launch of this engine may be in flight on any OTHER
 * stream while they run (it would see half-old, half-new tables) - join such streams into `stream` first.
 * Every adsp_set_spectrum* call forgets the kernel-reach hint (it described the previous kernel). */
ADSP_API int adsp_set_spectrum_device(adsp_engine* engine, const float* d_spectrum_interleaved, int n_bins, void* stream);

/* The one collective of the multi-GPU path: every engine of `engines[0..n)` - ONE engine per GPU, all
 * in this process, all with the geometry of engines[root] - takes over the filter of engines[root].  The root's
 * spectrum goes to its device, an RCCL broadcast (ncclBroadcast inside ncclGroupStart/End, one communicator per device
 * from ncclCommInitAll: a single process needs no rendezvous) carries it over xGMI, and every engine - the root
 * included - rebuilds its tables from what the collective left in ITS memory, so all n filters are bit-identical; the
 * kernel-reach hint travels with it.  librccl.so is opened on first use (an RCCL already in the process wins, then
 * $ADSP_RCCL_LIB, then the loader path); nothing else in this library needs it.  Set-up path: each device is drained.
 * Channels never interact (each reference device is private state, Example2.py:13-21), so this is the only collective;
 * a one-process-per-GPU host (torchrun) uses its own RCCL and adsp_set_spectrum_device instead. */
ADSP_API int adsp_bcast_spectrum(adsp_engine* const* engines, int n, int root);
/* The same collective for a ONE-PROCESS-PER-GPU host (round 4): this process holds rank `rank` of `world` and one engine.
 * Rank 0 draws 128 bytes with adsp_rccl_unique_id (ncclGetUniqueId) and hands them to every other rank by whatever means
 * the host has - a file, an environment variable, MPI, a socket; pyaudiodsptools_amd.dist uses a file next to the
 * rendezvous port - then EVERY rank calls adsp_bcast_spectrum_rank with the same id: the first call joins the communicator
 * (ncclCommInitRank: RCCL's own bootstrap, no torch.distributed; cached per id for later filter changes), a header
 * broadcast carries the root's geometry, sample format, spectrum precision and kernel-reach hint (a rank whose engine was
 * built for another window layout fails with ADSP_ERR_ARG instead of filtering with the wrong offsets), the spectrum
 * follows, and each rank rebuilds its tables from what the collective left in its own memory.  A float64 spectrum
 * (adsp_set_spectrum_f64) travels as float64 in both forms of the collective.  Set-up path: the device is drained.
 * The OUTCOME is collective as far as this library can make it: every rank enters both broadcasts whatever is wrong with its own
 * engine.  A root without a spectrum says so in the header and EVERY rank returns ADSP_ERR_STATE without entering the second
 * broadcast; a rank whose geometry differs still receives the spectrum (into scratch memory of the root's size, dropped) and only
 * then returns ADSP_ERR_ARG, so the other ranks complete.  A rank whose call fails BEFORE the first broadcast (NULL / out-of-range
 * arguments, no device, RCCL missing) or inside RCCL cannot be waited out by the others: treat any failure of this call on any
 * rank as fatal for the job (abort all ranks) - which is why the launcher-side checks (bench.py: ranks_seen, spectrum checksum)
 * exist.  Threads: the communicator table is locked only while it is read or written, never across ncclCommInitRank or the
 * broadcast, so several ranks may be driven from threads of one process.  adsp_rccl_finalize destroys every communicator the
 * library has built (no broadcast may be in flight); without it they live until the process ends. */
#define ADSP_RCCL_UNIQUE_ID_BYTES 128
ADSP_API int adsp_rccl_unique_id(char* unique_id /* [ADSP_RCCL_UNIQUE_ID_BYTES] */);
ADSP_API int adsp_bcast_spectrum_rank(adsp_engine* engine, const char* unique_id, int rank, int world, int root);
ADSP_API int adsp_rccl_finalize(void);
/* The spectrum this engine's tables were last built from, n_bins = F/2+1 interleaved (re, im) float32 values - after a
 * broadcast: what the collective left in this engine's device memory (bench.py checksums it across ranks). */
ADSP_API int adsp_get_spectrum(const adsp_engine* engine, float* spectrum_interleaved, int n_bins);
/* Version code of the RCCL that adsp_bcast_spectrum uses (ncclGetVersion); ADSP_ERR_STATE when none can be opened. */
ADSP_API int adsp_rccl_version(int* version);

/* 1 when the spectrum last set is real (every imaginary part exactly 0 - a symmetric kernel centred on circular index
 * 0): the kernel then runs its cheaper spectrum stage.  0 otherwise. */
ADSP_API int adsp_spectrum_is_real(const adsp_engine* engine, int* is_real);

/* Optional hint that saves input traffic: how many taps of the (circularly placed) kernel sit at NEGATIVE circular
 * indices - 0 for a causal kernel, (L-1)/2 for a zero-phase one centred on index 0.  Window positions at or beyond
 * out_offset + block_outputs + reach then feed discarded outputs only and are not fetched: a single-step launch of the
 * reference's cut filters reads 1.5 N instead of 2 N samples per chunk.  Negative = unknown (default): fetch everything. */
ADSP_API int adsp_set_kernel_reach(adsp_engine* engine, int taps_at_negative_indices);

/* Samples kept per transform in multi-step launches (apply_device with n_steps > 1).  Default is
 * chunk_size; any multiple of N/4 (specialised kernels; 4*threads_per_transform for the generic one) up to
 * fft_size - out_offset is valid and larger is cheaper (1.5 N for the cut filters at F = 2N). */
ADSP_API int adsp_set_block_outputs(adsp_engine* engine, int block_outputs);

/* Stateless effects, fused on the filter kernel's output registers (no extra HBM traffic) or run as a
 * standalone elementwise kernel.  Formulas and parameter meaning follow the reference:
 *   VOLUME           Utility.py:171-194 VolumeChange          p0 = 10^(dB/20), p1 != 0 clips to [-1, 1]
 *   SOFT_CLIPPER     EffectSoftClipper.py:23-45               p0 = drive + 1
 *   HARD_DISTORTION  EffectHardDistortion.py:17-41            (no parameters)
 *   SATURATOR        EffectSaturator.py:16-49                 p0 = 10^(threshold_dB/20), p1 = 10^(makeup_dB/20), p2 = 1 hard / 2 soft
 *   TREMOLO          EffectTremolo.py:19-57                   p0 = depth, p1 = lfo_hz / sampling_rate, p2 = LFO table length
 *                    (samples; the reference's len(arange(float32(fs / lfo)))).  Stateful only through the table index:
 *                    a fused tremolo starts at index 0 when it is set and advances with every chunk the engine filters,
 *                    following the reference's buffer arithmetic (EffectTremolo.py:40-45) including its replay quirk. */
#define ADSP_EFFECT_NONE 0
#define ADSP_EFFECT_VOLUME 1
#define ADSP_EFFECT_SOFT_CLIPPER 2
#define ADSP_EFFECT_HARD_DISTORTION 3
#define ADSP_EFFECT_SATURATOR 4
#define ADSP_EFFECT_TREMOLO 5
#define ADSP_EFFECT_BIT_CRUSHER 6 /* _EffectBitCrusher.py:8-12 (private in the reference): int16(trunc(32767 x)) // 512 / 64 */
/* every later apply of a float32 engine returns effect(filter(x)); ADSP_EFFECT_NONE removes it */
ADSP_API int adsp_set_epilogue(adsp_engine* engine, int effect, float p0, float p1, float p2);
/* out[i] = effect(in[i]) on device / host float32 arrays of n values (in-place allowed).  phase = LFO table index of
 * element 0 (tremolo only, 0 <= phase < p2; ignored by the other effects) */
ADSP_API int adsp_effect_device(int device_id, int effect, float p0, float p1, float p2, int phase, const float* d_in,
                                float* d_out, size_t n, void* stream);
ADSP_API int adsp_effect_host(int device_id, int effect, float p0, float p1, float p2, int phase, const float* in,
                              float* out, size_t n);
/* The tremolo over a [rows][row_len] device batch whose rows are the chunks of `rows` channels of ONE step: every row starts at LFO
 * table index `phase` (the reference runs one CreateTremolo per channel, all in step: EffectTremolo.py:27-47).  What engines that
 * cannot fuse the effect (one engine pass per kernel slice) run behind their last pass.  In place allowed. */
ADSP_API int adsp_tremolo_rows_device(int device_id, float depth, float lfo_per_sample, int lfo_length, int phase, const float* d_in,
                                      float* d_out, int rows, int row_len, void* stream);
/* MixSignals (Utility.py:51-72): out[i] = sum_j inputs[j][i], clipped to [-1, 1] when clip != 0.  `inputs` is a HOST
 * array of k device (adsp_mix_device) or host (adsp_mix_host) pointers; out may alias an input when k <= 8. */
ADSP_API int adsp_mix_device(int device_id, const float* const* d_inputs, int k, int clip, float* d_out, size_t n, void* stream);
ADSP_API int adsp_mix_host(int device_id, const float* const* inputs, int k, int clip, float* out, size_t n);

/* Non-finite inputs as the reference treats them.  The reference transforms chunks k-2, k-1, k as ONE 3N-point buffer
 * (EffectFFTFilter.py:67-72, EffectEQ3BandFFT.py:175-179), so one NaN / Inf sample turns the WHOLE returned chunk into NaN in
 * the call that takes it and in the two calls after it (tests/golden/kat_nonfinite.npz).  The engines' overlap-save blocks
 * poison only the blocks whose window holds the sample - a subset of those three chunks, a superset of the FIR's support.
 * This pass, launched on `stream` behind the filter call that turned d_in into d_out ([n_channels][chunk_size] float32 each),
 * restores the reference's behaviour per channel: it scans the new chunk, records the finding in slot `slot` (= call index
 * modulo 3) of the channel's flag ring d_flags[n_channels][3] (caller-owned device memory, zero before the first call) and
 * overwrites the channel's output chunk with NaN when any of its three slots is set.  The drop-in classes' apply() on
 * device-resident chunks uses it; host chunks are checked on the host. */
ADSP_API int adsp_nonfinite_guard(int device_id, const float* d_in, float* d_out, int n_channels, int chunk_size, unsigned* d_flags,
                                  int slot, void* stream);

/* Output mode of a float32 engine.  0: overwrite the output buffer (default).  1: ADD the filtered samples to what
 * the buffer holds - later parts of a partitioned convolution (a kernel longer than one transform is split into parts,
 * one engine per part, each with the part's taps and delay), or several engines summing onto one mix bus.
 * 2: add and clip the sum to [-1, 1] - the last engine of a MixSignals bus (Utility.py:51-72). */
ADSP_API int adsp_set_accumulate(adsp_engine* engine, int mode);

/* Forget all history (a fresh reference device). */
ADSP_API int adsp_reset(adsp_engine* engine);

/* The reference's apply() on host buffers, batched: in/out are [n_steps][C][N] host arrays
 * (in is only read, out is fully overwritten) of the engine's sample type.  H2D -> kernel -> D2H, synchronous. */
ADSP_API int adsp_apply_host(adsp_engine* engine, const void* in, void* out, int n_steps);

/* The measured path: device-resident [n_steps][C][N] batches, asynchronous on `stream`
 * (a hipStream_t passed as void*; NULL = the default stream).  d_in must stay unmodified until the
 * work on `stream` has finished.  History is carried inside the engine between calls. */
ADSP_API int adsp_apply_device(adsp_engine* engine, const void* d_in, void* d_out, int n_steps, void* stream);

/* Zero-copy streaming: the producer (H2D copy, decoder, generator kernel) writes the next chunk
 * batch [C][N] straight into the ring slot returned by adsp_ring_acquire, then adsp_apply_ring
 * filters it.  No state copy, 1.25-1.75 N reads + N writes per channel per step.
 * Consecutive steps are independent kernels (step k reads ring slots k-history .. k and writes its own output):
 * issuing step k's producer + adsp_apply_ring on stream k % 2 of TWO streams lets the next launch fill the CUs the
 * previous one is draining.  The steps still depend on each other THROUGH THE RING, and the library orders them:
 *   - step k reads the slots of steps k-1 .. k-history, filled by producers on other streams: adsp_apply_ring makes its
 *     stream wait for an event recorded when each of those steps was submitted (their producers were complete);
 *   - the producer of step k overwrites the slot of step k - ring_slots, which the kernels of steps k - ring_slots ..
 *     k - ring_slots + history have read: adsp_ring_acquire_stream makes the producer's stream wait for those kernels.
 * Nothing is recorded while every step arrives on one stream; the first step on a different stream joins the old stream
 * once, and from then on each step costs two event records and up to history_chunks + 1 stream waits.  Producers must be
 * enqueued on the stream given to adsp_ring_acquire_stream.  adsp_ring_acquire_stream is MANDATORY for every step that is
 * not issued on the previous step's stream - the first one on a new stream included: plain adsp_ring_acquire knows no
 * stream, orders nothing while the library has seen a single stream (the stream itself orders that case), and only once
 * several streams have been seen blocks the HOST until the slot's last readers have finished.  More slots than
 * history_chunks + 1 let a producer run further ahead of the kernels; correctness does not depend on the count. */
ADSP_API int adsp_ring_acquire(adsp_engine* engine, void** d_slot);
ADSP_API int adsp_ring_acquire_stream(adsp_engine* engine, void** d_slot, void* stream);
ADSP_API int adsp_apply_ring(adsp_engine* engine, void* d_out, void* stream);
/* The same overlap without the caller juggling streams (round 4): adsp_ring_set_pipeline(engine, 2) makes the LIBRARY run step k
 * on its own stream k % 2.  The caller keeps ONE stream: adsp_ring_acquire_stream(engine, &slot, stream) orders its producer after
 * the kernels that still read the slot, the producer is enqueued on `stream`, adsp_apply_ring(engine, d_out, stream) records an
 * event on `stream` (everything enqueued so far = the producer), lets the step's own stream wait for it and launches there; the
 * per-step events above order the steps through the ring.  What changes for the caller: the outputs are NOT ordered on `stream`
 * any more - adsp_ring_join(engine, stream) makes `stream` wait for every step issued so far (call it before `stream` reads
 * d_out, or synchronise the device).  Needs ring_slots >= history_chunks + 2; depth 1 restores the default.  Measured on
 * config 2's per-chunk pattern (4096 channels x 4096 samples, one launch per chunk): 44 us per step against 48.6 on one stream.
 * Depth 3 (round 5): the same three calls RIDE A LIVE SESSION (below) that the library starts on first use, feeds and stops by itself -
 * one persistent launch with the history on chip instead of a launch per step (config 3, 4096 channels x 512 samples: the step a
 * real-time caller pays drops from ~8 us to the session's ~5.6 us without a second API).  adsp_ring_acquire[_stream] returns the next
 * slot once the session has left it (the host waits where it lags), adsp_apply_ring enqueues ONE one-lane kernel on `stream` behind the
 * producer - it stores the step's output address and publishes the step - and returns; adsp_ring_join BLOCKS THE HOST until every step
 * submitted so far has its outputs in memory (they are written through: any stream may read them afterwards).  Refused with
 * ADSP_ERR_ARG where no session can hold the engine (more channel groups than the GPU keeps resident of that kernel, generic geometry,
 * int16, fused effect): fall back to depth 2.  A session that sees no step for the time-out of adsp_live_configure (default 1 s) ends
 * by itself and the next step starts a fresh one; any call that needs the engine in its ordinary state (adsp_apply_device, the
 * spectrum setters, adsp_reset, a depth change ..) winds the session down first - after every step submitted so far has been consumed.  The outputs of step k must not be read before a
 * join that follows its adsp_apply_ring. */
ADSP_API int adsp_ring_set_pipeline(adsp_engine* engine, int depth);
ADSP_API int adsp_ring_join(adsp_engine* engine, void* stream);
/* Resident ring launches: ONE launch consumes the next n_steps ring steps, and the workgroups of step k start as soon
 * as the producer side has PUBLISHED that step - consecutive steps overlap inside one grid, with no launch boundary
 * between them (per-step launches of small chunk batches spend a third of each step ramping up and draining: config 3,
 * 4096 channels x 512 samples, EffectEQ3BandFFT.py:19).  The consumer may be launched before its input exists.
 *   producer side, ONE stream, in step order:
 *     adsp_ring_produce_begin(engine, &d_slot, stream)   slot of the next step to fill; orders `stream` after the
 *                                                         launches that still read the slot's old contents
 *     ... enqueue the copy / kernel that fills d_slot [C][N] on `stream` ...
 *     adsp_ring_produce_end(engine, stream)              enqueues the publication of EVERY slot handed out since the last
 *                                                         one: a 32-bit sequence word in device memory advances once
 *                                                         everything before it on `stream` is done (a publication costs
 *                                                         the host ~10 us: publish per step when steps arrive one by
 *                                                         one, per batch when the producer fills several slots in a row)
 *   consumer side:
 *     adsp_apply_ring_resident(engine, d_out, n_steps, stream)   d_out [n_steps][C][N]; 1 <= n_steps <= ring_slots -
 *                                                         history_chunks (size the ring for the run-ahead wanted).
 *   Each workgroup polls the word for its step (thread 0, system-scope loads, s_sleep between polls), every wave then
 *   executes an acquire fence before it loads the window.  A workgroup that has waited longer than the time-out
 *   (default 250 ms, adsp_ring_resident_timeout) gives up WITHOUT writing its outputs and raises a flag that
 *   adsp_ring_resident_status returns (and clears): a consumer without a producer cannot hang the GPU.  After a
 *   time-out the stream state is undefined: adsp_reset.
 *   Waiting workgroups HOLD their CU slots (the launch runs step-major, so only the steps next in line are resident).
 *   LIMITATION (measured, tools/resident_probe.py, profiles/r3_resident.txt): a launch whose grid is larger than the GPU
 *   holds at once (n_steps x workgroups per step > ~3000) and whose NEXT step is not yet published stops dispatching, and on
 *   MI355X / ROCm 7.2 the producer stream's commands - the runtime's own write of the sequence word included - then do not
 *   get through either: the launch ends by time-out.  Launch the consumer before its input only with grids the GPU can hold
 *   (the -m gpu test: 20 workgroups per step); at full size publish first - the launch then never waits and runs at the
 *   multi-step rate (config 3: 5.5-6.1 us per step against 8.0 for one launch per step).
 *   Use explicitly created streams for both sides: work on the legacy default (NULL) stream is implicitly ordered against
 *   every blocking stream, so a producer would wait for the very launch that waits for it.
 *   Resident and per-step ring calls do not mix: adsp_ring_reset_order (a drain) switches between them.
 *   Not available for the generic-geometry kernels and with a fused tremolo. */
ADSP_API int adsp_ring_produce_begin(adsp_engine* engine, void** d_slot, void* stream);
ADSP_API int adsp_ring_produce_end(adsp_engine* engine, void* stream);
ADSP_API int adsp_apply_ring_resident(adsp_engine* engine, void* d_out, int n_steps, void* stream);
ADSP_API int adsp_ring_resident_timeout(adsp_engine* engine, double milliseconds);
ADSP_API int adsp_ring_resident_status(adsp_engine* engine, int* timed_out);

/* Live sessions (round 4): the real-time call pattern of the reference (a chunk arrives, apply() runs, the next chunk arrives:
 * Example3.py:20-34; config 3 = CreateEQ3BandFFT on 2048 stereo pairs of 512-sample chunks, EffectEQ3BandFFT.py:156-211) as ONE
 * persistent launch.  adsp_live_start launches it on `stream`: one workgroup per channel group plus a relay workgroup, ALL
 * resident at once (refused with ADSP_ERR_ARG when the engine has more channel groups than the GPU holds of this kernel), each
 * looping over the steps of its channel group.  A step is consumed as soon as it is PUBLISHED; its N outputs per channel go to
 * slot (step % out_slots) of d_out [out_slots][C][N].  The history the next window needs stays in registers, so every input
 * sample is read from memory once (8 bytes of traffic per sample); the new chunk is read with system-scope loads and the
 * outputs are written through, so the session may outlive any number of ring laps and readers on other streams / the host see
 * the outputs once the progress count says so.
 *   producer, per step (one host thread):
 *     adsp_live_slot(engine, &d_slot)        ring slot [C][N] of the next step; ADSP_ERR_STATE "ring full" while the session is
 *                                            ring_slots - history_chunks steps behind (poll adsp_live_progress / adsp_live_wait)
 *     ... fill d_slot ...
 *     adsp_live_publish_host(engine)         the data is complete and visible (blocking copy, or the producer's stream was
 *                                            synchronised): ONE plain store to a host-mapped word - no HIP call, no command on
 *                                            any queue; the relay workgroup forwards it to the device word the workers poll
 *     adsp_live_publish_stream(engine, s)    or: a one-lane kernel enqueued on stream s behind the commands that fill the slot
 *                                            bumps the device word itself (device-side producers may also do that from their
 *                                            own kernels: adsp_live_device_words)
 *   consumer side: adsp_live_progress (steps every channel group has completed, outputs in memory; a host-mapped word, no HIP
 *   call), adsp_live_wait (spins on it), adsp_live_stop (ends the session once every published step is consumed, synchronises
 *   `stream`, advances the engine's ring by the steps consumed - per-step calls may follow).
 * A workgroup that waits longer than the step time-out (adsp_live_configure, default 1000 ms, 0 = for ever) gives up and the
 * session ends; adsp_live_stop then returns ADSP_ERR_STATE.  stream = NULL (recommended) runs the session on a stream of the
 * library's own, created with the HIGHEST priority: the launch never ends while its producer lives, and every command that
 * follows it in the same HARDWARE queue waits for it - HIP maps streams onto a handful of hardware queues (on MI355X / ROCm
 * 7.2 every sixth stream created shared the NULL stream's queue; a producer's copy then sat behind the session until the
 * session timed out), and streams of another priority come from another pool of queues.  A caller-provided stream must be
 * non-blocking and must not share a hardware queue with any stream that feeds the session.  Available for float32 engines in the stream geometry (power-of-two chunk 128 ..
 * 4096, fft_size = 2 x chunk_size) with lookback 5/4 N (the cut filters) or 7/4 N (the 3-band EQ); no fused effect.
 * Output ring: step s overwrites slot s % out_slots of d_out whether or not anybody has read step s - out_slots; only the INPUT
 * ring has flow control (adsp_live_slot).  The consumer keeps up to within out_slots steps of the producer - e.g. by reading step s
 * before it publishes step s + out_slots, which a producer that waits for adsp_live_progress >= s + 1 before handing out slot
 * s + out_slots does by construction - or sizes out_slots for the whole session.
 * While a session runs the engine's other entry points that would touch the ring or the kernel's configuration are refused with
 * ADSP_ERR_STATE (per-step, multi-step and resident calls, adsp_ring_produce_*, adsp_set_accumulate, adsp_set_epilogue, the
 * spectrum setters, reset / state calls): adsp_live_stop first.
 * While a session runs, anything that drains the whole device waits for it: hipDeviceSynchronize, and the set-up calls of this
 * library that contain one (adsp_set_spectrum, adsp_reset, adsp_get_state / adsp_set_state, adsp_destroy of ANY engine on the
 * device) - stop the session first, or do the set-up before it starts.
 * load_mode: how the new chunk is read - 2 system scope (default), 1 non-temporal, 0 plain (tuning A/B only: a plain load may
 * hit a cache line of an earlier ring lap). */
ADSP_API int adsp_live_configure(adsp_engine* engine, double step_timeout_ms, int load_mode);
ADSP_API int adsp_live_start(adsp_engine* engine, void* d_out, int out_slots, unsigned max_steps, void* stream);
ADSP_API int adsp_live_slot(adsp_engine* engine, void** d_slot);
ADSP_API int adsp_live_publish_host(adsp_engine* engine);
ADSP_API int adsp_live_publish_stream(adsp_engine* engine, void* stream);
/* benchmark / soak helper: a data-less producer in a native loop - takes and publishes the next n_steps slots one by one
 * (whatever they hold is the input), through a one-lane kernel per step on `stream` (use_stream != 0) or through host stores,
 * waiting for ring space where the session lags */
ADSP_API int adsp_live_publish_run(adsp_engine* engine, unsigned n_steps, int use_stream, void* stream);
ADSP_API int adsp_live_progress(adsp_engine* engine, unsigned* steps_done);
ADSP_API int adsp_live_wait(adsp_engine* engine, unsigned steps, double timeout_ms);
ADSP_API int adsp_live_device_words(adsp_engine* engine, unsigned** d_published, unsigned** d_done);
ADSP_API int adsp_live_stop(adsp_engine* engine, unsigned* steps_consumed);

/* Drain the device and forget the per-step ordering events.  Needed around hipGraph capture of ring steps: events
 * recorded inside a capture must not be waited on outside it (and vice versa), so call this before the capture begins
 * and again after it ends.  adsp_reset / adsp_get_state do the same as a side effect. */
ADSP_API int adsp_ring_reset_order(adsp_engine* engine);

/* Test hooks: the engine's history as [history_chunks][C][N] host floats, oldest first
 * (the reference's float32_array_input_3/_2). */
ADSP_API int adsp_get_state(adsp_engine* engine, void* host_history);
/* The state a fused effect adds to a checkpoint: for the tremolo the position of its LFO (the length of the reference's
 * table buffer, EffectTremolo.py:40-45); 0 for every stateless effect.  adsp_reset restarts the LFO as well. */
ADSP_API int adsp_get_epilogue_state(const adsp_engine* engine, long long* state);
ADSP_API int adsp_set_epilogue_state(adsp_engine* engine, long long state);
ADSP_API int adsp_set_state(adsp_engine* engine, const void* host_history);

/* Kernel timing for benchmarks: when enabled, every launch of the filter kernel is bracketed by a pair
 * of HIP events recorded on the launch stream (the kernel only - not the history copy that follows a
 * multi-step launch).  adsp_kernel_time synchronises those events, returns the summed kernel time in
 * milliseconds and the number of launches since the last call, and clears the record. */
ADSP_API int adsp_enable_kernel_timing(adsp_engine* engine, int enable);
ADSP_API int adsp_kernel_time(adsp_engine* engine, double* total_ms, int* launches);

/* Shader clock under load (bench.py, SURVEY 8d: the chip clocks to its power budget, so a throughput figure is only
 * comparable together with the clock it ran at): adsp_clock_probe_launch enqueues a one-lane kernel on `stream` - a side
 * stream next to the timed launches - that counts shader cycles (s_memtime) over `microseconds` of the constant 100 MHz
 * clock; adsp_clock_probe_read waits for `stream`, returns cycles / time in MHz and frees `result`. */
ADSP_API int adsp_clock_probe_launch(int device_id, double microseconds, void* stream, unsigned long long** result);
ADSP_API int adsp_clock_probe_read(int device_id, void* stream, unsigned long long* result, double* shader_mhz);

/* Block until everything this engine enqueued on `stream` is done. */
ADSP_API int adsp_synchronize(adsp_engine* engine, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Tapped delay line: the arithmetic of the reference's CreateDelay (EffectDelay.py:60-72) and of its
 * reverb's delay lines (_EffectReverb.py:46-58), both of which feed the output of the FFT filters into it:
 *     out[t] = dry_gain * x[t] + sum_k tap_gain[k] * x[t - tap_delay[k]]          (per channel, zero initial history)
 * CreateDelay(time T samples, L loops): tap_delay = T, 2T, .. LT; tap_gain = linspace(0.5, 0.1, L); dry 1 (0 if wet).
 * float32 [step][channel][sample] batches like the FFT engines, so the two chain on the device without reshaping.
 * ------------------------------------------------------------------------------------------------------------- */
#define ADSP_DELAY_MAX_TAPS 1024
typedef struct adsp_delay adsp_delay; /* opaque */
typedef struct adsp_delay_config {
    int device_id;
    int chunk_size; /* N: any multiple of 4 */
    int n_channels;
    int n_taps;     /* 0..ADSP_DELAY_MAX_TAPS */
} adsp_delay_config;
/* tap_delay[k] >= 1 samples, any order; the engine keeps ceil(max delay / N) chunks of input history on the device */
ADSP_API int adsp_delay_create(const adsp_delay_config* cfg, const int* tap_delay, const float* tap_gain, float dry_gain,
                               adsp_delay** out);
ADSP_API void adsp_delay_destroy(adsp_delay* line);
ADSP_API int adsp_delay_reset(adsp_delay* line); /* history back to zeros */
/* non-zero: add the result to what the output buffer holds (the reverb sums two lines) */
ADSP_API int adsp_delay_set_accumulate(adsp_delay* line, int on);
ADSP_API int adsp_delay_history_chunks(const adsp_delay* line, int* chunks);
/* d_in / d_out: device [n_steps][n_channels][chunk_size] float32, NOT aliased; asynchronous on `stream` */
ADSP_API int adsp_delay_apply_device(adsp_delay* line, const float* d_in, float* d_out, int n_steps, void* stream);
/* host buffers, synchronous (with accumulate on, `out` is read as well as written) */
ADSP_API int adsp_delay_apply_host(adsp_delay* line, const float* in, float* out, int n_steps);

/* ---------------------------------------------------------------------------------------------------------------
 * Per-channel sequential scans (last item): the reference's recursive devices.  One lane per channel,
 * float32 [step][channel][sample] batches like everything else, state carried across calls, in-place allowed.
 *   biquad cascade - EffectEQ3Band.py:95-181: every section is
 *        y[i] = f32( c0 x[i-1] + c1 x[i-2] + c2 x[i-3] - c3 y[i-1] - c4 y[i-2] )        (float64, left to right)
 *     with c = (b0, b1, b2, a1, a2) / a0; the one-sample input delay is the reference's (it prepends three old inputs
 *     but two old outputs).  Sections run in series (applylowband -> applymidband -> applyhighband = 3 sections).
 *   compressor - EffectCompressor.py:43-125: attack / hold / release state machine over two gain envelopes
 *     (linspace(1, ratio, attack samples), linspace(ratio, 1, release samples)), threshold on |x|.
 *   gate - EffectGate.py:42-126: the same state machine over linspace(1, 1/depth) / linspace(1/depth, 1); the threshold
 *     is tested on the raw sample and the sample is scaled by `depth` before the envelope: out = (x * depth) * env.
 * ------------------------------------------------------------------------------------------------------------- */
#define ADSP_SCAN_BIQUAD 1
#define ADSP_SCAN_COMPRESSOR 2
#define ADSP_SCAN_GATE 3
#define ADSP_SCAN_MAX_SECTIONS 4
typedef struct adsp_scan adsp_scan; /* opaque */
typedef struct adsp_scan_config {
    int device_id;
    int chunk_size;  /* any positive length */
    int n_channels;
    int kind;        /* set by the create functions */
    int n_sections;  /* biquad cascade: 1..ADSP_SCAN_MAX_SECTIONS */
} adsp_scan_config;
/* coefficients: [n_sections][5] doubles b0/a0, b1/a0, b2/a0, a1/a0, a2/a0 */
ADSP_API int adsp_scan_create_biquad(const adsp_scan_config* cfg, const double* coefficients, adsp_scan** out);
ADSP_API int adsp_scan_create_compressor(const adsp_scan_config* cfg, float threshold, const float* attack_envelope,
                                         int n_attack, const float* release_envelope, int n_release, adsp_scan** out);
ADSP_API int adsp_scan_create_gate(const adsp_scan_config* cfg, float threshold, float depth, const float* attack_envelope,
                                   int n_attack, const float* release_envelope, int n_release, adsp_scan** out);
ADSP_API void adsp_scan_destroy(adsp_scan* scan);
ADSP_API int adsp_scan_reset(adsp_scan* scan);
ADSP_API int adsp_scan_apply_device(adsp_scan* scan, const float* d_in, float* d_out, int n_steps, void* stream);
ADSP_API int adsp_scan_apply_host(adsp_scan* scan, const float* in, float* out, int n_steps);

/* ---------------------------------------------------------------------------------------------------------------
 * Exact mode: the streaming FIR as a float64 DIRECT sum,  out[tau] = sum_t taps[t] * s[tau - delay - t]  per channel
 * (zero history), for the reference's WAV front end where the export truncates: with ADSP_FORMAT_S16 the
 * conversions are the reference's to the letter - float32(pcm)/32768 in (Utility.py:236-237), then
 * int16(trunc(float32(y) * 32767)) out (EffectFFTFilter.py:75, Utility.py:306) - and the result is the int16 stream the
 * reference writes, bit for bit wherever its own complex128 pipeline is within 1e-9 of the exact value.  O(taps) per
 * sample: meant for files, not for the batched hot path.  With ADSP_FORMAT_F32 it returns float32(y): the on-device
 * ground truth the parity tests compare every channel of the FFT engines with.
 * For one reference device: taps = the L-tap kernel, delay = chunk_size - (L-1)/2 (one chunk of latency minus the
 * look-ahead d of the kept slice, EffectFFTFilter.py:22-25).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct adsp_exact adsp_exact; /* opaque */
typedef struct adsp_exact_config {
    int device_id;
    int chunk_size;    /* N: any positive length */
    int n_channels;    /* 1..65535 */
    int n_taps;
    int delay;         /* >= 0 samples */
    int sample_format; /* ADSP_FORMAT_F32 or ADSP_FORMAT_S16: type of the in/out batches */
} adsp_exact_config;
ADSP_API int adsp_exact_create(const adsp_exact_config* cfg, const double* taps, adsp_exact** out);
ADSP_API void adsp_exact_destroy(adsp_exact* fir);
ADSP_API int adsp_exact_reset(adsp_exact* fir); /* history back to zeros */
/* d_in / d_out: device [n_steps][n_channels][chunk_size], NOT aliased; asynchronous on `stream` */
ADSP_API int adsp_exact_apply_device(adsp_exact* fir, const void* d_in, void* d_out, int n_steps, void* stream);
ADSP_API int adsp_exact_apply_host(adsp_exact* fir, const void* in, void* out, int n_steps);

/* ---------------------------------------------------------------------------------------------------------------
 * Uniformly partitioned engines (round 5): streaming FIRs LONGER than one transform - the reference's own GPU example runs
 * chunk_size 88200 (Example4.py:5, ModuleTestsGPU.py:35: CreateLowCutFilter -> 44 099 taps, CreateEQ3BandFFT -> 88 197).
 *     out[tau] = y[tau - delay],   y = taps (*) s   (zero history),   taps cut into P partitions of B taps, B one of
 *     adsp_upols_block_sizes() (8192, 16384: per output sample the second launch reads P x 16 bytes - 8 of table, 8 of spectrum)
 * Every input block of B samples is transformed ONCE (2B-point real FFT), its spectrum kept in a frequency-domain delay line in HBM;
 * an output block is ONE inverse transform of sum_p X_{b-p} H_p.  Two launches per call (forward transforms; multiply-accumulate +
 * inverse + store), each over every (channel, block) at once - instead of one full engine pass per kernel slice.
 *   spectra : [n_partitions][B + 1] interleaved (re, im) float32 = rfft(partition p of the taps zero-padded to 2B), computed by the
 *             host in float64.  int16 engines: the host folds 32767/32768 into the taps (like ADSP_FORMAT_S16 engines).
 *   delay   : multiple of 4, >= B (a reference device: chunk_size - look-ahead, rounded down to a multiple of 4 by delaying the
 *             kernel by 0..3 taps).  chunk_size: multiple of 4, >= 16.  Batches: [step][channel][sample] like everything else.
 *   max_steps: chunks one pair of launches covers at most (longer calls are split); sizes the delay line:
 *             (ceil((max_steps * chunk_size + delay) / B) + n_partitions + 3) blocks of 8 B bytes per channel.
 * A fused effect (ADSP_EFFECT_*; float32 engines) is applied to the output registers; the tremolo's LFO follows the stream's own time
 * base (table index 0 at the first sample after adsp_upols_set_epilogue / adsp_upols_reset, the reference's buffer quirk included).
 * Calls of one engine are ordered by the library whatever streams they are given (a call on another stream than the previous one
 * first waits for an event recorded behind the previous call's launches); one host thread per engine at a time.
 * After every call the engine keeps only the LAST 2B samples of the input it was given (no later window reaches further back).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct adsp_upols adsp_upols; /* opaque */
typedef struct adsp_upols_config {
    int device_id;
    int chunk_size;
    int n_channels;
    int block_size;     /* one of adsp_upols_block_sizes() */
    int n_partitions;
    int delay;
    int sample_format;  /* ADSP_FORMAT_F32 or ADSP_FORMAT_S16 */
    int max_steps;
} adsp_upols_config;
ADSP_API int adsp_upols_block_size(void);                       /* the smallest block size of this build */
ADSP_API int adsp_upols_block_sizes(int* sizes, int capacity);  /* ascending; returns how many there are (also with sizes == NULL) */
ADSP_API int adsp_upols_create(const adsp_upols_config* cfg, const float* spectra, adsp_upols** out);
ADSP_API void adsp_upols_destroy(adsp_upols* fir);
ADSP_API int adsp_upols_reset(adsp_upols* fir); /* history and delay line back to zeros */
ADSP_API int adsp_upols_set_epilogue(adsp_upols* fir, int effect, float p0, float p1, float p2);
ADSP_API int adsp_upols_info(const adsp_upols* fir, int* history_chunks, int* delay_line_blocks, size_t* delay_line_bytes);
/* d_in / d_out: device [n_steps][n_channels][chunk_size], NOT aliased; asynchronous on `stream` */
ADSP_API int adsp_upols_apply_device(adsp_upols* fir, const void* d_in, void* d_out, int n_steps, void* stream);
ADSP_API int adsp_upols_apply_host(adsp_upols* fir, const void* in, void* out, int n_steps);
/* The filter of a running engine (same partitioning: n_partitions x (block + 1) interleaved bins): replace it (set-up path: drains
 * the device), read back what the tables were last built from, or take it over from another engine / rank - SURVEY 8e's ONE collective
 * for kernels longer than a transform.  adsp_upols_bcast_spectra: one process, one engine per GPU (like adsp_bcast_spectrum);
 * adsp_upols_bcast_spectra_rank: one process per GPU, `unique_id` from adsp_rccl_unique_id on rank 0 (like adsp_bcast_spectrum_rank).
 * The payload carries (chunk, block, partitions, delay, format): an engine partitioned differently keeps its own filter and fails. */
ADSP_API int adsp_upols_set_spectra(adsp_upols* fir, const float* spectra);
ADSP_API int adsp_upols_get_spectra(const adsp_upols* fir, float* spectra, size_t n_floats);
ADSP_API int adsp_upols_bcast_spectra(adsp_upols* const* engines, int n, int root);
ADSP_API int adsp_upols_bcast_spectra_rank(adsp_upols* fir, const char* unique_id, int rank, int world, int root);
/* wait until everything this engine has launched - on `stream` or, if its last call went elsewhere, there - has finished */
ADSP_API int adsp_upols_synchronize(adsp_upols* fir, void* stream);
/* The block of the convolution axis that straddles the end of a call can be computed ONCE - its second part carried, as float32, to the head of
 * the next call's output by that call's per-channel workgroups - or in both calls (rounds 5 - 6b).  Same samples either way; carrying saves one
 * multiply + inverse transform per channel and call (one block in 6.4 at blocks of 16384 and Example4's chunk: -5 ... -7 % per call from 128
 * channels on) and costs the short calls of few channels 1 - 2 % (one more copy on the call's critical path).  mode: -1 = the library decides
 * per call (carry when the multiply launch has at least two workgroups per CU; the default), 0 = never, 1 = always (tests). */
ADSP_API int adsp_upols_set_carry(adsp_upols* fir, int mode);
/* Checkpoint / resume (the reference's whole state is its two previous chunks, EffectFFTFilter.py:40-42 - a partitioned
 * engine's is the transformed past): get_state drains the device and copies counters, input ring and frequency-domain delay line
 * into `state` (adsp_upols_state_bytes bytes: ~8 B per sample of delay line, 1.75 MiB per channel for Example4's low cut); set_state
 * on an engine of the same configuration continues the stream bit for bit.  A state of another shape is refused. */
ADSP_API int adsp_upols_state_bytes(const adsp_upols* fir, size_t* bytes);
ADSP_API int adsp_upols_get_state(adsp_upols* fir, void* state, size_t capacity);
ADSP_API int adsp_upols_set_state(adsp_upols* fir, const void* state, size_t bytes);

/* ---------------------------------------------------------------------------------------------------------------
 * Counter-based synthetic input: sample (channel c, absolute index t) is a pure function of (seed, c, t) -
 * uniform(-1, 1) float32 (2^24 equidistant values, scaled by `amplitude`) or uniform int16 in [-16384, 16384) - written on the
 * device straight into a [n_steps][n_channels][chunk_size] batch whose first sample of channel j is absolute index first_sample
 * of channel first_channel + j.  pyaudiodsptools_amd/synth.py holds the bit-identical numpy twin, so a host-side checker can
 * regenerate any channel of a resident batch (bench.py checks its timed output against the CPU oracle that way).
 * chunk_size must be a multiple of 4.  Asynchronous on `stream`.
 * ------------------------------------------------------------------------------------------------------------- */
ADSP_API int adsp_synth_device(int device_id, unsigned seed, unsigned first_channel, unsigned long long first_sample, int n_channels,
                               int chunk_size, int n_steps, int sample_format, float amplitude, void* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADSP_H */
