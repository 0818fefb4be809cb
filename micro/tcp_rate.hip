// micro/tcp_rate.hip - vector-memory (TA/TCP) issue-rate microbenchmark for gfx950 (tuning aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// each wave sweeps a window of `win_bytes` (per wave) repeatedly with coalesced loads of W floats per lane
template <int W, bool STRIDED>
__global__ void k(const float* __restrict__ buf, float* out, int iters, int win_floats) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const float* base = buf + (size_t)(wave % 64) * win_floats;  // 64 distinct windows, shared by many waves (L2/L1 hits)
    float acc = 0.f;
    const int per_instr = STRIDED ? 64 * 32 : 64 * W;  // floats covered per wave-instruction
    int off = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* p = base + off + (STRIDED ? lane * 32 : lane * W);
            if constexpr (W == 1) { acc += *p; }
            else if constexpr (W == 2) { float2 v = *(const float2*)p; acc += v.x + v.y; }
            else { float4 v = *(const float4*)p; acc += v.x + v.y + v.z + v.w; }
            off += per_instr; if (off + per_instr > win_floats) off = 0;
        }
    }
    if (acc == 1234.5f) out[0] = acc;
}

template <int W, bool STRIDED>
int run(const char* name, const float* d, float* o, int win_floats, int waves_per_simd) {
    const int wgs = 256 * waves_per_simd, thr = 256, iters = 2048;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<W, STRIDED>), dim3(wgs), dim3(thr), 0, 0, d, o, 64, win_floats);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<W, STRIDED>), dim3(wgs), dim3(thr), 0, 0, d, o, iters, win_floats);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = (double)wgs * thr / 64 / 256 * iters * 8;
    printf("%-34s win=%7d B occ=%d  %8.3f ms  %7.2f ns/wave-instr/CU  %8.1f GB/s/CU  %7.2f TB/s chip\n", name, win_floats * 4, waves_per_simd, ms,
           ms * 1e6 / instr_per_cu, instr_per_cu * 64 * W * 4 / (ms * 1e6), instr_per_cu * 64 * W * 4 / (ms * 1e6) * 256 / 1e3);
    return 0;
}

int main() {
    float *d, *o; const size_t n = 64 * (1 << 20) / 4;  // 64 MiB
    CHECK(hipMalloc(&d, n * 4)); CHECK(hipMalloc(&o, 1024)); CHECK(hipMemset(d, 0, n * 4));
    for (int occ : {2, 8}) {
        for (int win : {2048, 65536}) {  // floats per window: 8 KiB (L1-resident) / 256 KiB (L2-resident)
            run<1, false>("dword   coalesced", d, o, win, occ);
            run<2, false>("dwordx2 coalesced", d, o, win, occ);
            run<4, false>("dwordx4 coalesced", d, o, win, occ);
            run<2, true>("dwordx2 lane-stride 128B", d, o, win, occ);
            run<4, true>("dwordx4 lane-stride 128B", d, o, win, occ);
        }
    }
    return 0;
}
