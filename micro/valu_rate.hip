// micro/valu_rate.hip - VALU issue-rate microbenchmark for gfx950 (tuning aid, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void k(float* out, int iters, float seed) {
    float a[16]; v2f p[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed * (i + threadIdx.x); p[i] = v2f{a[i], a[i] + 1.f}; }
    float c = seed * 1.0001f; v2f pc = v2f{c, c * 0.5f};
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {  // 16 independent v_fma_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(c));
        } else if constexpr (MODE == 1) {  // 16 independent v_pk_fma_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(pc));
        } else if constexpr (MODE == 2) {  // 16 v_add_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if constexpr (MODE == 3) {  // 16 v_pk_add_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
        } else if constexpr (MODE == 4) {  // 16 v_pk_mul_f32 with op_sel swizzle + neg
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(p[i]) : "v"(pc));
        } else if constexpr (MODE == 5) {  // dependent chain v_fma_f32 (1 accumulator)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[0]) : "v"(c));
        } else if constexpr (MODE == 6) {  // dependent chain v_pk_fma_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[0]) : "v"(pc));
        } else if constexpr (MODE == 7) {  // 8 v_fma + 8 s_add (mixed scalar/vector issue)
            int s = it;
#pragma unroll
            for (int i = 0; i < 8; ++i) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(c)); asm volatile("s_add_i32 %0, %0, 1" : "+s"(s)); }
            if (s == -1) a[0] += 1.f;
        } else if constexpr (MODE == 8) {  // v_mov_b32 x16
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(c));
        } else if constexpr (MODE == 9) {  // v_pk_mov_b32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_mov_b32 %0, %1, %1" : "=v"(p[i]) : "v"(pc));
        }
    }
    float r = 0; for (int i = 0; i < 16; ++i) r += a[i] + p[i].x + p[i].y;
    if (r == 12345.678f) out[0] = r;
}

template <int MODE>
int run(const char* name, int wgs, int threads, float* d) {
    const int iters = 4096;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(threads), 0, 0, d, 16, 0.5f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(threads), 0, 0, d, iters, 0.5f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    // instructions per SIMD: waves per SIMD * iters * 16
    const double waves = (double)wgs * threads / 64.0, per_simd = waves / 1024.0 * iters * 16;
    printf("%-28s wgs=%5d thr=%4d waves/SIMD=%5.2f  %8.3f ms  -> %.2f ns per wave-instr per SIMD (x clk GHz = cycles)\n", name, wgs, threads,
           waves / 1024.0, ms, ms * 1e6 / per_simd);
    return 0;
}

int main() {
    float* d; CHECK(hipMalloc(&d, 1024));
    for (int occ : {1, 2, 4, 8}) {
        const int wgs = 256 * occ, thr = 256;  // occ waves per SIMD
        run<0>("v_fma_f32 x16 indep", wgs, thr, d);
        run<1>("v_pk_fma_f32 x16 indep", wgs, thr, d);
        run<2>("v_add_f32 x16 indep", wgs, thr, d);
        run<3>("v_pk_add_f32 x16 indep", wgs, thr, d);
        run<4>("v_pk_mul_f32 opsel/neg x16", wgs, thr, d);
        run<5>("v_fma_f32 dependent", wgs, thr, d);
        run<6>("v_pk_fma_f32 dependent", wgs, thr, d);
        run<7>("8 v_fma + 8 s_add", wgs, thr, d);
        run<8>("v_mov_b32 x16", wgs, thr, d);
        run<9>("v_pk_mov_b32 x16", wgs, thr, d);
    }
    return 0;
}
