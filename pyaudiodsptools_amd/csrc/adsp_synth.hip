// adsp_synth.hip - counter-based synthetic input: uniform(-1, 1) float32 samples (or int16 PCM) that are a pure
// function of (seed, channel, absolute sample index), generated on the device straight into [step][channel][sample] batches.
// bench.py fills its resident batches with it, so that any channel of the TIMED input can be regenerated on the host
// (pyaudiodsptools_amd/synth.py: the numpy twin, bit-identical) and the timed output checked on the CPU by the test infrastructure - which a
// stateful generator (torch's Philox stream) does not allow without copying gigabytes back.
//
// The hash is 32-bit integer arithmetic only (so the twin is four numpy lines):
//   h = lo(t) * 0x9E3779B1 + hi(t) * 0x85EBCA77 + channel * 0xC2B2AE3D + seed * 0x27D4EB2F        (mod 2^32)
//   h = fmix32(h)                                                (MurmurHash3's finaliser: full avalanche)
//   float32: (h >> 8) * 2^-23 - 1        exact in float32, 2^24 equidistant values in [-1, 1)
//   int16  : (int16)(h >> 17) - 16384    uniform in [-16384, 16384): -6 dBFS PCM
#include <hip/hip_runtime.h>

#include "../../include/adsp.h"
#include "capi_common.hpp"

using adsp::fail;

namespace {

__device__ __forceinline__ unsigned synth_hash(unsigned seed, unsigned channel, unsigned long long t) {
    unsigned h = static_cast<unsigned>(t) * 0x9E3779B1u + static_cast<unsigned>(t >> 32) * 0x85EBCA77u + channel * 0xC2B2AE3Du + seed * 0x27D4EB2Fu;
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

// one thread = four consecutive samples of one channel (16-byte / 8-byte stores); grid-stride over [steps][C][N / 4]
template <bool S16>
__global__ __launch_bounds__(256) void synth_kernel(void* __restrict__ out, unsigned seed, unsigned channel0, unsigned long long sample0, int C, int N,
                                                    long long n_steps, float amplitude) {
    const long long quads_per_chunk = N / 4;
    const long long total = n_steps * C * quads_per_chunk;
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long q = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; q < total; q += stride) {
        const long long chunk = q / quads_per_chunk;          // step * C + channel
        const int i = static_cast<int>(q - chunk * quads_per_chunk) * 4;
        const long long step = chunk / C;
        const unsigned c = static_cast<unsigned>(chunk - step * C);
        const unsigned long long t = sample0 + static_cast<unsigned long long>(step) * static_cast<unsigned long long>(N) + static_cast<unsigned long long>(i);
        unsigned h[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = synth_hash(seed, channel0 + c, t + j);
        if constexpr (S16) {
            short v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = static_cast<short>(static_cast<int>(h[j] >> 17) - 16384);
            typedef short v4s __attribute__((ext_vector_type(4)));
            *reinterpret_cast<v4s*>(static_cast<short*>(out) + chunk * N + i) = v4s{v[0], v[1], v[2], v[3]};
        } else {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (static_cast<float>(h[j] >> 8) * 1.1920928955078125e-07f - 1.0f) * amplitude;
            *reinterpret_cast<float4*>(static_cast<float*>(out) + chunk * N + i) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

}  // namespace

extern "C" int adsp_synth_device(int device_id, unsigned seed, unsigned first_channel, unsigned long long first_sample, int n_channels, int chunk_size,
                                 int n_steps, int sample_format, float amplitude, void* d_out, void* stream) {
    if (!d_out) return fail(ADSP_ERR_ARG, "d_out is NULL");
    if (n_channels <= 0 || n_steps <= 0) return fail(ADSP_ERR_ARG, "n_channels and n_steps must be positive");
    if (chunk_size < 4 || chunk_size % 4) return fail(ADSP_ERR_ARG, "chunk_size %d: the generator writes four samples per lane, need a multiple of 4", chunk_size);
    if (sample_format != ADSP_FORMAT_F32 && sample_format != ADSP_FORMAT_S16)
        return fail(ADSP_ERR_ARG, "sample_format %d: ADSP_FORMAT_F32 or ADSP_FORMAT_S16 (int16 buffers of either int16 engine kind)", sample_format);
    HIP_TRY(hipSetDevice(device_id));
    const long long total = static_cast<long long>(n_steps) * n_channels * (chunk_size / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 workgroups per CU
    if (sample_format == ADSP_FORMAT_S16)
        hipLaunchKernelGGL(synth_kernel<true>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), d_out, seed, first_channel,
                           first_sample, n_channels, chunk_size, static_cast<long long>(n_steps), amplitude);
    else
        hipLaunchKernelGGL(synth_kernel<false>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), d_out, seed, first_channel,
                           first_sample, n_channels, chunk_size, static_cast<long long>(n_steps), amplitude);
    HIP_TRY(hipGetLastError());
    return ADSP_OK;
}
