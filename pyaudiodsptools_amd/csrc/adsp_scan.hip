// adsp_scan.hip - the reference's recursive devices (last item): per-channel sequential scans.
//
//   * biquad cascade   EffectEQ3Band.py:95-181   y[i] = f32(c0 x[i-1] + c1 x[i-2] + c2 x[i-3] - c3 y[i-1] - c4 y[i-2])
//   * compressor       EffectCompressor.py:43-125 attack / hold / release state machine over two gain envelopes
//   * gate             EffectGate.py:42-126       the same state machine; the threshold is tested on the raw sample, the
//                                                 sample is scaled by `depth` first and the envelopes run 1 <-> 1/depth
//
// Both carry state from one sample to the next, so time cannot be split across lanes without changing the rounding the
// reference's loops produce.  What is parallel is the channel axis: ONE LANE PER CHANNEL, 64 channels per workgroup (the biquads always; the
// compressor and the gate from a few thousand channels on - below that compressor_wave_kernel puts time across the lanes, see there).
// Memory stays coalesced through an LDS tile: the wave loads 64 channels x 64 samples row by row (256 contiguous bytes
// per row), each lane then walks its own row (rows padded to 65 floats: conflict-free), and the tile is stored back row
// by row.  Latency-bound by construction (a dependent chain of ~10 float64 operations per sample and section); the
// numbers are in DESIGN.md section 6f.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/adsp.h"
#include "capi_common.hpp"

using adsp::fail;

namespace {

constexpr int TILE = 64;

struct ScanArgs {
    const float* in;
    float* out;
    int C, N, n_steps;
    // biquad
    int n_sections;
    const double* coef;  // [n_sections][5]: b0/a0, b1/a0, b2/a0, a1/a0, a2/a0
    float* bq_state;     // [n_sections][5][C]: x[-3], x[-2], x[-1], y[-2], y[-1]
    // compressor
    float threshold;
    float pre_gain;  // gate: the sample is scaled by `depth` before the envelope (EffectGate.py:59); compressor: 1
    const float* attack;
    const float* release;
    int x_max, y_max;
    int* cp_state;  // [3][C]: x, y, state (0 resting, 1 attack, 2 release)
    int env_in_lds;  // the two envelopes fit the dynamic LDS block: the per-sample gain lookup never leaves the CU
};

template <int NS>  // sections: compile-time so that coefficients and state live in registers
struct Biquad {
    double c[NS][5];
    float xh[NS][3], yh[NS][2];
    static constexpr int ns = NS;
    __device__ void load(const ScanArgs& a, int ch) {
#pragma unroll
        for (int s = 0; s < ns; ++s) {
            for (int k = 0; k < 5; ++k) c[s][k] = a.coef[s * 5 + k];
            for (int k = 0; k < 3; ++k) xh[s][k] = a.bq_state[(static_cast<size_t>(s) * 5 + k) * a.C + ch];
            for (int k = 0; k < 2; ++k) yh[s][k] = a.bq_state[(static_cast<size_t>(s) * 5 + 3 + k) * a.C + ch];
        }
    }
    __device__ void save(const ScanArgs& a, int ch) const {
#pragma unroll
        for (int s = 0; s < ns; ++s) {
            for (int k = 0; k < 3; ++k) a.bq_state[(static_cast<size_t>(s) * 5 + k) * a.C + ch] = xh[s][k];
            for (int k = 0; k < 2; ++k) a.bq_state[(static_cast<size_t>(s) * 5 + 3 + k) * a.C + ch] = yh[s][k];
        }
    }
    __device__ void stage_tables(const ScanArgs&, float*, int) {}
    __device__ void begin_chunk() {}
    // float64, left to right, every product and sum rounded on its own (no fma) - what numpy's scalar arithmetic does
    __device__ float sample(float v) {
#pragma clang fp contract(off)
#pragma unroll
        for (int s = 0; s < ns; ++s) {
            const double acc = c[s][0] * static_cast<double>(xh[s][2]) + c[s][1] * static_cast<double>(xh[s][1]) +
                               c[s][2] * static_cast<double>(xh[s][0]) - c[s][3] * static_cast<double>(yh[s][1]) -
                               c[s][4] * static_cast<double>(yh[s][0]);
            const float y = static_cast<float>(acc);
            xh[s][0] = xh[s][1];
            xh[s][1] = xh[s][2];
            xh[s][2] = v;  // this sample is used from the next output on (the reference's one-sample input delay)
            yh[s][0] = yh[s][1];
            yh[s][1] = y;
            v = y;
        }
        return v;
    }
};

template <bool ENV_LDS>  // gain envelopes staged in LDS (typed pointer: ds_read, not a flat load) or left in global memory
struct Compressor {
    int x, y, state;
    bool full, freeze;
    int where;  // 0 top, 1 attack, 2 hold, 3 release - position in the reference's loop nest, restarts at every chunk
    float threshold, pre_gain;
    const float *attack, *release;
    int x_max, y_max;
    double ratio;
    __device__ void load(const ScanArgs& a, int ch) {
        x = a.cp_state[ch];
        y = a.cp_state[a.C + ch];
        state = a.cp_state[2 * a.C + ch];
        threshold = a.threshold;
        pre_gain = a.pre_gain;
        if constexpr (!ENV_LDS) {
            attack = a.attack;
            release = a.release;
        }
        x_max = a.x_max;
        y_max = a.y_max;
        ratio = static_cast<double>(x_max) / static_cast<double>(y_max);  // the reference's x_max / y_max, once
    }
    // every lane of the workgroup: copy the gain envelopes next to the tile (a gain lookup per sample from HBM/L2
    // would put a memory round trip into the dependent chain)
    __device__ void stage_tables(const ScanArgs& a, float* lds, int lane) {
        if constexpr (ENV_LDS) {
            for (int i = lane; i < a.x_max; i += TILE) lds[i] = a.attack[i];
            for (int i = lane; i < a.y_max; i += TILE) lds[a.x_max + i] = a.release[i];
            attack = lds;
            release = lds + a.x_max;
        }
    }
    __device__ void save(const ScanArgs& a, int ch) const {
        a.cp_state[ch] = x;
        a.cp_state[a.C + ch] = y;
        a.cp_state[2 * a.C + ch] = state;
    }
    __device__ void begin_chunk() {
        full = true;
        freeze = false;
        where = 0;
    }
    // One sample in, one sample out: the loop nest of EffectCompressor.py:67-124 advanced until it consumes the sample.
    // At most seven positions are visited per sample (a chunk that starts inside a release with a loud sample: top ->
    // attack -> hold -> release, interrupted -> top -> attack -> hold), so the walk
    // is a fixed, predicated sequence: lanes of a wave sit at different positions and a data-dependent loop with
    // early returns serialises them.  A sample that passes untouched is multiplied by 1.0f (exact).
    __device__ float sample(float v) {
        const bool above = fabsf(v) > threshold;
        float gain = 1.0f;
        bool done = false;
        for (int visit = 0; visit < 8; ++visit) {
            if (!__any(!done)) break;  // the whole wave has consumed its sample (usually after one or two visits)
            if (done) continue;
            if (where == 0) {
                if (above || x != 0 || y != 0) {
                    if (full && state == 0) {
                        x = 0;
                        state = 1;
                    }
                    if (!full && state == 2) {
                        x = x_max - static_cast<int>(static_cast<double>(y) * ratio);
                        freeze = false;
                        state = 1;
                    }
                    where = 1;
                } else {
                    done = true;
                }
            } else if (where == 1) {
                if (x < x_max && state == 1) {
                    gain = attack[x++];
                    done = true;
                } else {
                    where = 2;
                }
            } else if (where == 2) {
                if (above && state == 1) {
                    gain = attack[x_max - 1];
                    done = true;
                } else {
                    state = 2;
                    where = 3;
                }
            } else {
                bool consumed = false;
                if (y < y_max && state == 2) {
                    x = 0;
                    if (!above) {
                        gain = release[y++];
                        consumed = true;
                    } else {
                        full = false;
                        y = 0;
                        freeze = true;
                    }
                }
                if (consumed) {
                    done = true;
                } else {
                    if (y == y_max) {
                        full = true;
                        state = 0;
                        x = 0;
                        y = 0;
                    }
                    where = 0;
                    done = !freeze;  // the sample after a completed release passes untouched
                }
            }
        }
        // gate: (x * depth) * envelope, two float32 products like the reference's array multiply followed by the
        // in-place element multiply; compressor: pre_gain = 1 (exact).  An untouched sample is multiplied by 1.0f.
        return (v * pre_gain) * gain;
    }
};

// One wave = CH channels (4, 16 or 64: the host picks the smallest that still gives the chip a thousand waves - the
// recurrence is a dependent chain of hundreds of cycles per sample, so what counts is how many SIMDs have a wave at all:
// 4096 channels are 64 waves at 64 channels per wave and 1024 waves at 4, +4x measured).  Time is walked in tiles of 64
// samples: while the first CH lanes run the recurrence over tile t (each its own LDS row), the CH row loads of tile t+1 are
// already in flight into registers - the loads are the only latency a lone wave can hide.
template <class Op, int CH>
__global__ __launch_bounds__(TILE) void scan_kernel(const ScanArgs a) {
    __shared__ float tile[CH][TILE + 1];
    const int lane = static_cast<int>(threadIdx.x);
    const int c0 = static_cast<int>(blockIdx.x) * CH;
    const int ch = c0 + lane;
    const bool mine = lane < CH && ch < a.C;
    const int rows = a.C - c0 < CH ? a.C - c0 : CH;
    const int tiles_per_chunk = (a.N + TILE - 1) / TILE;
    const int n_tiles = a.n_steps * tiles_per_chunk;
    extern __shared__ float dyn_lds[];
    Op op;
    if (mine) op.load(a, ch);
    op.stage_tables(a, dyn_lds, lane);  // visible after the first __syncthreads below

    // tile index -> (step, first sample); element (row r, column lane) of a tile
    auto tile_base = [&](int t, int& w) -> size_t {
        const int s = t / tiles_per_chunk, t0 = (t - s * tiles_per_chunk) * TILE;
        w = a.N - t0 < TILE ? a.N - t0 : TILE;
        return (static_cast<size_t>(s) * a.C + c0) * a.N + t0;
    };
    float pre[CH];
    auto fetch = [&](int t) {
        int w;
        const size_t base = tile_base(t, w);
#pragma unroll
        for (int r = 0; r < CH; ++r) pre[r] = (r < rows && lane < w) ? a.in[base + static_cast<size_t>(r) * a.N + lane] : 0.f;
    };
    fetch(0);
    for (int t = 0; t < n_tiles; ++t) {
        int w;
        const size_t base = tile_base(t, w);
#pragma unroll
        for (int r = 0; r < CH; ++r) tile[r][lane] = pre[r];
        __syncthreads();
        if (t + 1 < n_tiles) fetch(t + 1);  // in flight during the recurrence below
        if (mine) {
            if (t % tiles_per_chunk == 0) op.begin_chunk();
            for (int k = 0; k < w; ++k) tile[lane][k] = op.sample(tile[lane][k]);
        }
        __syncthreads();
        if (lane < w)
            for (int r = 0; r < rows; ++r) a.out[base + static_cast<size_t>(r) * a.N + lane] = tile[r][lane];
        __syncthreads();
    }
    if (mine) op.save(a, ch);
}

// Compressor / gate with TIME across the lanes (round 6): one wave = one channel, its 64 lanes = 64 consecutive samples.  What is sequential in
// the reference's loop nest is only WHICH gain a sample gets - the walk reads the threshold bit of every sample and three counters, never
// a sample value.  So the wave takes the 64 threshold bits with one ballot and walks the state machine over them in wave-uniform integer
// code (no LDS or memory access inside the chain), and then all 64 lanes look their gain up and multiply at once.  Walking sample by sample
// that is ~100 scalar instructions per sample at one instruction per four cycles (what ONE wave issues, scalar or vector): 1.5x faster
// than a lane walking its LDS row, no more.  The walk therefore moves in RUNS: at its four steady positions (resting, attack ramp, hold,
// release ramp) the length of the run is a count of trailing zeros / ones of the threshold bits, and the lanes of the run take their
// gain codes in one step - a sine through the default compressor is ~12 steps per 64 samples, a signal below the threshold one.
// Same transitions, same tables, same two float32 products as Compressor::sample - bit for bit (tests/test_gpu_recursive.py holds the two
// kernels against each other).  Used up to kWaveScanMaxChannels channels (2048 with envelopes above 16 KiB): beyond that one lane per
// channel (scan_kernel above) has more channels in flight than one wave per channel can walk.
constexpr int kWaveScanWaves = 4;             // channels per workgroup (they share the staged envelopes)
// measured (profiles/r6f_scan_time_across_lanes.txt): the compressor's envelopes (6 KiB) leave the CU full of workgroups and the form wins up
// to 4096 channels (1.4x there, 2.5x at 1024; -8 % at 8192); the gate's (35 KiB: four workgroups per CU) from 4096 channels on one lane per channel is ahead
constexpr int kWaveScanMaxChannels = 4096, kWaveScanMaxChannelsBigEnvelopes = 2048;

template <bool ENV_LDS>
__global__ __launch_bounds__(TILE * kWaveScanWaves) void compressor_wave_kernel(const ScanArgs a) {
    extern __shared__ float dyn_lds[];
    const int lane = static_cast<int>(threadIdx.x) & (TILE - 1);
    const int ch = static_cast<int>(blockIdx.x) * kWaveScanWaves + (static_cast<int>(threadIdx.x) >> 6);
    const float* table = nullptr;
    if constexpr (ENV_LDS) {
        for (int i = static_cast<int>(threadIdx.x); i < a.x_max; i += TILE * kWaveScanWaves) dyn_lds[i] = a.attack[i];
        for (int i = static_cast<int>(threadIdx.x); i < a.y_max; i += TILE * kWaveScanWaves) dyn_lds[a.x_max + i] = a.release[i];
        __syncthreads();
        table = dyn_lds;  // attack[0 .. x_max), release[0 .. y_max) behind it: gain code = index into this
    }
    if (ch >= a.C) return;  // (wave-uniform; after the only workgroup barrier)
    const int x_max = a.x_max, y_max = a.y_max;
    int x = __builtin_amdgcn_readfirstlane(a.cp_state[ch]);
    int y = __builtin_amdgcn_readfirstlane(a.cp_state[a.C + ch]);
    int state = __builtin_amdgcn_readfirstlane(a.cp_state[2 * a.C + ch]);
    const double ratio = static_cast<double>(x_max) / static_cast<double>(y_max);
    const float threshold = a.threshold, pre_gain = a.pre_gain;
    const int tiles_per_chunk = (a.N + TILE - 1) / TILE;
    const int n_tiles = a.n_steps * tiles_per_chunk;
    bool full = true, freeze = false;
    int where = 0;

    auto tile_at = [&](int t, int& w) -> size_t {
        const int s = t / tiles_per_chunk, t0 = (t - s * tiles_per_chunk) * TILE;
        w = a.N - t0 < TILE ? a.N - t0 : TILE;
        return (static_cast<size_t>(s) * a.C + ch) * a.N + t0;
    };
    int w0;
    const size_t base0 = tile_at(0, w0);
    float pre = lane < w0 ? a.in[base0 + lane] : 0.f;
    for (int t = 0; t < n_tiles; ++t) {
        int w;
        const size_t base = tile_at(t, w);
        const float v = pre;
        if (t + 1 < n_tiles) {  // in flight during the walk below
            int wn;
            const size_t bn = tile_at(t + 1, wn);
            pre = lane < wn ? a.in[bn + lane] : 0.f;
        }
        if (t % tiles_per_chunk == 0) {  // the loop nest restarts with every chunk (Compressor::begin_chunk)
            full = true;
            freeze = false;
            where = 0;
        }
        const unsigned long long above_bits = __ballot(lane < w && fabsf(v) > threshold);
        int code = -1;  // this lane's gain: -1 untouched (x 1.0f), else index into attack ++ release
        int k = 0;
        while (k < w) {
            // Steady positions of the walk: a RUN of samples that Compressor::sample would consume one by one without moving - resting
            // until the next sample above the threshold, the attack ramp (whatever the samples are), the hold while they stay above, the
            // release ramp while they stay below.  One step per run: its length from the threshold bits (count of trailing zeros / ones),
            // the lanes of the run take their codes at once.  Everything else - the transitions - goes through the walk below.
            const unsigned long long rest = above_bits >> k;  // bit 0 = sample k; zero beyond the tile
            int n = 0, first = -1, stride = 0;
            if (where == 0 && x == 0 && y == 0) {
                n = rest ? __builtin_ctzll(rest) : TILE;
            } else if (where == 1 && state == 1 && x < x_max) {
                n = x_max - x;
                first = x;
                stride = 1;
            } else if (where == 2 && state == 1) {
                n = ~rest ? __builtin_ctzll(~rest) : TILE;
                first = x_max - 1;
            } else if (where == 3 && state == 2 && y < y_max) {
                n = rest ? __builtin_ctzll(rest) : TILE;
                n = n < y_max - y ? n : y_max - y;
                first = x_max + y;
                stride = 1;
            }
            n = n < w - k ? n : w - k;
            if (n > 0) {
                if (first >= 0 && lane >= k && lane < k + n) code = first + (lane - k) * stride;
                if (where == 1) x += n;
                if (where == 3) {
                    y += n;
                    x = 0;
                }
                k += n;
                continue;
            }
            const bool above = rest & 1ull;
            int c = -1;
            bool done = false;
            for (int visit = 0; visit < 8 && !done; ++visit) {  // Compressor::sample's walk, on wave-uniform values
                if (where == 0) {
                    if (above || x != 0 || y != 0) {
                        if (full && state == 0) {
                            x = 0;
                            state = 1;
                        }
                        if (!full && state == 2) {
                            x = x_max - __builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<double>(y) * ratio));
                            freeze = false;
                            state = 1;
                        }
                        where = 1;
                    } else {
                        done = true;
                    }
                } else if (where == 1) {
                    if (x < x_max && state == 1) {
                        c = x++;
                        done = true;
                    } else {
                        where = 2;
                    }
                } else if (where == 2) {
                    if (above && state == 1) {
                        c = x_max - 1;
                        done = true;
                    } else {
                        state = 2;
                        where = 3;
                    }
                } else {
                    bool consumed = false;
                    if (y < y_max && state == 2) {
                        x = 0;
                        if (!above) {
                            c = x_max + y++;
                            consumed = true;
                        } else {
                            full = false;
                            y = 0;
                            freeze = true;
                        }
                    }
                    if (consumed) {
                        done = true;
                    } else {
                        if (y == y_max) {
                            full = true;
                            state = 0;
                            x = 0;
                            y = 0;
                        }
                        where = 0;
                        done = !freeze;
                    }
                }
            }
            code = lane == k ? c : code;
            ++k;
        }
        float gain = 1.0f;
        if (code >= 0) {
            if constexpr (ENV_LDS) gain = table[code];
            else gain = code < x_max ? a.attack[code] : a.release[code - x_max];
        }
        if (lane < w) a.out[base + lane] = (v * pre_gain) * gain;
    }
    if (lane == 0) {
        a.cp_state[ch] = x;
        a.cp_state[a.C + ch] = y;
        a.cp_state[2 * a.C + ch] = state;
    }
}

template <class Op>
void launch_scan(const ScanArgs& a, size_t dyn_lds, hipStream_t stream) {
    // channels per wave: as few as it takes to put >= 1024 waves on the chip (256 CUs x 4 SIMDs)
    const int ch = a.C >= 64 * 1024 ? 64 : a.C >= 16 * 1024 ? 16 : 4;
    const unsigned grid = static_cast<unsigned>((a.C + ch - 1) / ch);
    if (ch == 64)
        hipLaunchKernelGGL((scan_kernel<Op, 64>), dim3(grid), dim3(TILE), dyn_lds, stream, a);
    else if (ch == 16)
        hipLaunchKernelGGL((scan_kernel<Op, 16>), dim3(grid), dim3(TILE), dyn_lds, stream, a);
    else
        hipLaunchKernelGGL((scan_kernel<Op, 4>), dim3(grid), dim3(TILE), dyn_lds, stream, a);
}

}  // namespace

struct adsp_scan {
    adsp_scan_config cfg;
    double* d_coef;
    float* d_bq_state;
    float *d_attack, *d_release;
    int x_max, y_max;
    float threshold;
    float pre_gain;
    int* d_cp_state;
    float* stage;
    size_t stage_elems;
};

namespace {
int common_checks(const adsp_scan_config* cfg, adsp_scan** out) {
    if (!cfg || !out) return fail(ADSP_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (cfg->chunk_size < 1) return fail(ADSP_ERR_ARG, "chunk_size must be positive");
    if (cfg->n_channels < 1) return fail(ADSP_ERR_ARG, "n_channels must be positive");
    int ndev = 0;
    int rc = adsp_device_count(&ndev);
    if (rc) return rc;
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(ADSP_ERR_ARG, "device_id %d out of range (%d devices)", cfg->device_id, ndev);
    HIP_TRY(hipSetDevice(cfg->device_id));
    return ADSP_OK;
}
adsp_scan* blank(const adsp_scan_config* cfg) {
    adsp_scan* e = new adsp_scan();
    e->cfg = *cfg;
    e->d_coef = nullptr;
    e->d_bq_state = nullptr;
    e->d_attack = e->d_release = nullptr;
    e->d_cp_state = nullptr;
    e->stage = nullptr;
    e->stage_elems = 0;
    e->x_max = e->y_max = 0;
    e->threshold = 0.f;
    e->pre_gain = 1.f;
    return e;
}
}  // namespace

extern "C" {

int adsp_scan_create_biquad(const adsp_scan_config* cfg, const double* coefficients, adsp_scan** out) {
    int rc = common_checks(cfg, out);
    if (rc) return rc;
    if (!coefficients) return fail(ADSP_ERR_ARG, "NULL coefficients");
    if (cfg->n_sections < 1 || cfg->n_sections > ADSP_SCAN_MAX_SECTIONS) return fail(ADSP_ERR_ARG, "n_sections %d: need 1..%d", cfg->n_sections, ADSP_SCAN_MAX_SECTIONS);
    adsp_scan* e = blank(cfg);
    e->cfg.kind = ADSP_SCAN_BIQUAD;
    auto bail = [&](int code) {
        adsp_scan_destroy(e);
        return code;
    };
    const size_t nstate = (size_t)cfg->n_sections * 5 * cfg->n_channels;
    hipError_t err;
    if ((err = hipMalloc(&e->d_coef, (size_t)cfg->n_sections * 5 * sizeof(double))) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMalloc: %s", hipGetErrorString(err)));
    if ((err = hipMemcpy(e->d_coef, coefficients, (size_t)cfg->n_sections * 5 * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(err)));
    if ((err = hipMalloc(&e->d_bq_state, nstate * sizeof(float))) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMalloc: %s", hipGetErrorString(err)));
    if ((err = hipMemset(e->d_bq_state, 0, nstate * sizeof(float))) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMemset: %s", hipGetErrorString(err)));
    *out = e;
    return ADSP_OK;
}

int adsp_scan_create_compressor(const adsp_scan_config* cfg, float threshold, const float* attack_envelope, int n_attack,
                                const float* release_envelope, int n_release, adsp_scan** out) {
    int rc = common_checks(cfg, out);
    if (rc) return rc;
    if (!attack_envelope || !release_envelope) return fail(ADSP_ERR_ARG, "NULL envelope");
    if (n_attack < 1 || n_release < 1) return fail(ADSP_ERR_ARG, "envelopes need at least one sample each (the reference indexes attack[len - 1])");
    adsp_scan* e = blank(cfg);
    e->cfg.kind = ADSP_SCAN_COMPRESSOR;
    e->threshold = threshold;
    e->x_max = n_attack;
    e->y_max = n_release;
    auto bail = [&](int code) {
        adsp_scan_destroy(e);
        return code;
    };
    hipError_t err;
    if ((err = hipMalloc(&e->d_attack, n_attack * sizeof(float))) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMalloc: %s", hipGetErrorString(err)));
    if ((err = hipMalloc(&e->d_release, n_release * sizeof(float))) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMalloc: %s", hipGetErrorString(err)));
    if ((err = hipMemcpy(e->d_attack, attack_envelope, n_attack * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(err)));
    if ((err = hipMemcpy(e->d_release, release_envelope, n_release * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(err)));
    if ((err = hipMalloc(&e->d_cp_state, (size_t)3 * cfg->n_channels * sizeof(int))) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMalloc: %s", hipGetErrorString(err)));
    if ((err = hipMemset(e->d_cp_state, 0, (size_t)3 * cfg->n_channels * sizeof(int))) != hipSuccess) return bail(fail(ADSP_ERR_HIP, "hipMemset: %s", hipGetErrorString(err)));
    *out = e;
    return ADSP_OK;
}

int adsp_scan_create_gate(const adsp_scan_config* cfg, float threshold, float depth, const float* attack_envelope,
                          int n_attack, const float* release_envelope, int n_release, adsp_scan** out) {
    if (!(depth > 0.f)) return fail(ADSP_ERR_ARG, "gate depth must be positive (the envelopes run to 1/depth)");
    int rc = adsp_scan_create_compressor(cfg, threshold, attack_envelope, n_attack, release_envelope, n_release, out);
    if (rc) return rc;
    (*out)->cfg.kind = ADSP_SCAN_GATE;
    (*out)->pre_gain = depth;
    return ADSP_OK;
}

void adsp_scan_destroy(adsp_scan* e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device_id);
    (void)hipDeviceSynchronize();
    for (void* p : {(void*)e->d_coef, (void*)e->d_bq_state, (void*)e->d_attack, (void*)e->d_release, (void*)e->d_cp_state, (void*)e->stage})
        if (p) (void)hipFree(p);
    delete e;
}

int adsp_scan_reset(adsp_scan* e) {
    if (!e) return fail(ADSP_ERR_ARG, "NULL engine");
    HIP_TRY(hipSetDevice(e->cfg.device_id));
    HIP_TRY(hipDeviceSynchronize());
    if (e->d_bq_state) HIP_TRY(hipMemset(e->d_bq_state, 0, (size_t)e->cfg.n_sections * 5 * e->cfg.n_channels * sizeof(float)));
    if (e->d_cp_state) HIP_TRY(hipMemset(e->d_cp_state, 0, (size_t)3 * e->cfg.n_channels * sizeof(int)));
    return ADSP_OK;
}

int adsp_scan_apply_device(adsp_scan* e, const float* d_in, float* d_out, int n_steps, void* stream) {
    if (!e || !d_in || !d_out) return fail(ADSP_ERR_ARG, "NULL argument");
    if (n_steps <= 0) return fail(ADSP_ERR_ARG, "n_steps must be positive");
    HIP_TRY(hipSetDevice(e->cfg.device_id));
    ScanArgs a{};
    a.in = d_in;
    a.out = d_out;
    a.C = e->cfg.n_channels;
    a.N = e->cfg.chunk_size;
    a.n_steps = n_steps;
    a.n_sections = e->cfg.n_sections;
    a.coef = e->d_coef;
    a.bq_state = e->d_bq_state;
    a.threshold = e->threshold;
    a.pre_gain = e->pre_gain;
    a.attack = e->d_attack;
    a.release = e->d_release;
    a.x_max = e->x_max;
    a.y_max = e->y_max;
    a.cp_state = e->d_cp_state;
    const size_t env_bytes = ((size_t)e->x_max + (size_t)e->y_max) * sizeof(float);
    a.env_in_lds = (e->cfg.kind != ADSP_SCAN_BIQUAD && env_bytes <= 40 * 1024) ? 1 : 0;
    if (e->cfg.kind == ADSP_SCAN_BIQUAD) {
        switch (e->cfg.n_sections) {
            case 1: launch_scan<Biquad<1>>(a, 0, (hipStream_t)stream); break;
            case 2: launch_scan<Biquad<2>>(a, 0, (hipStream_t)stream); break;
            case 3: launch_scan<Biquad<3>>(a, 0, (hipStream_t)stream); break;
            default: launch_scan<Biquad<4>>(a, 0, (hipStream_t)stream); break;
        }
    } else if (a.C <= (env_bytes > 16 * 1024 ? kWaveScanMaxChannelsBigEnvelopes : kWaveScanMaxChannels) && !getenv("ADSP_SCAN_LANE_PER_CHANNEL")) {  // time across the lanes (the variable: A/B and tests of the other kernel)
        const unsigned grid = static_cast<unsigned>((a.C + kWaveScanWaves - 1) / kWaveScanWaves);
        if (a.env_in_lds)
            hipLaunchKernelGGL(compressor_wave_kernel<true>, dim3(grid), dim3(TILE * kWaveScanWaves), env_bytes, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL(compressor_wave_kernel<false>, dim3(grid), dim3(TILE * kWaveScanWaves), 0, (hipStream_t)stream, a);
    } else if (a.env_in_lds)
        launch_scan<Compressor<true>>(a, env_bytes, (hipStream_t)stream);
    else
        launch_scan<Compressor<false>>(a, 0, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return ADSP_OK;
}

int adsp_scan_apply_host(adsp_scan* e, const float* in, float* out, int n_steps) {
    if (!e || !in || !out) return fail(ADSP_ERR_ARG, "NULL argument");
    if (n_steps <= 0) return fail(ADSP_ERR_ARG, "n_steps must be positive");
    HIP_TRY(hipSetDevice(e->cfg.device_id));
    const size_t elems = (size_t)n_steps * e->cfg.n_channels * e->cfg.chunk_size;
    if (elems * sizeof(float) <= adsp::kHostWindowMax) {  // a chunk or a few: the kernel reads / writes pinned host memory (capi_common.hpp)
        adsp::HostWindow* w = adsp::host_window(e->cfg.device_id);
        if (!w) return ADSP_ERR_ARG;
        std::lock_guard<std::mutex> lock(w->mu);
        int rc = adsp::host_window_reserve(*w, elems * sizeof(float), elems * sizeof(float));
        if (rc) return rc;
        memcpy(w->in, in, elems * sizeof(float));
        if ((rc = adsp_scan_apply_device(e, static_cast<const float*>(w->d_in), static_cast<float*>(w->d_out), n_steps, nullptr))) return rc;
        if ((rc = adsp::host_window_wait(*w, nullptr))) return rc;
        memcpy(out, w->out, elems * sizeof(float));
        return ADSP_OK;
    }
    if (elems > e->stage_elems) {
        HIP_TRY(hipDeviceSynchronize());
        if (e->stage) (void)hipFree(e->stage);
        e->stage = nullptr;
        e->stage_elems = 0;
        HIP_TRY(hipMalloc(&e->stage, elems * sizeof(float)));
        e->stage_elems = elems;
    }
    HIP_TRY(hipMemcpy(e->stage, in, elems * sizeof(float), hipMemcpyHostToDevice));
    int rc = adsp_scan_apply_device(e, e->stage, e->stage, n_steps, nullptr);  // in place
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, e->stage, elems * sizeof(float), hipMemcpyDeviceToHost));
    return ADSP_OK;
}

}  // extern "C"
