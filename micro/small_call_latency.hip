// micro/small_call_latency.hip - what a small host call costs on MI355X, by completion mechanism.  A "call" = memcpy 2 KiB into a pinned, mapped
// buffer, one elementwise kernel over it (reads / writes the host memory over PCIe), wait, memcpy 2 KiB out.
//   A  hipEventRecord + hipEventSynchronize (what the library's small host calls do)
//   B  hipStreamSynchronize
//   C  a second one-lane kernel that writes a sequence number to mapped host memory; the host polls it
//   D  the elementwise kernel writes the sequence number itself (one workgroup: after a workgroup barrier and a system-scope fence)
// build: hipcc -O2 --offload-arch=gfx950 -o micro/small_call_latency micro/small_call_latency.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                                      \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

__global__ void work(const float* in, float* out, int n) {
    for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += gridDim.x * blockDim.x) out[i] = in[i] * 0.5f + 0.25f;
}
__global__ void flag(volatile unsigned* f, unsigned seq) { __hip_atomic_store(const_cast<unsigned*>(f), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void work_and_flag(const float* in, float* out, int n, unsigned* f, unsigned seq) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = in[i] * 0.5f + 0.25f;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(f, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int main() {
    const int n = 512, calls = 20000;
    float *hin, *hout;
    unsigned* hflag;
    CK(hipHostMalloc((void**)&hin, 65536, hipHostMallocMapped));
    CK(hipHostMalloc((void**)&hout, 65536, hipHostMallocMapped));
    CK(hipHostMalloc((void**)&hflag, 64, hipHostMallocMapped));
    float *din, *dout;
    unsigned* dflag;
    CK(hipHostGetDevicePointer((void**)&din, hin, 0));
    CK(hipHostGetDevicePointer((void**)&dout, hout, 0));
    CK(hipHostGetDevicePointer((void**)&dflag, hflag, 0));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    std::vector<float> x(n, 1.0f), y(n);
    *hflag = 0;
    unsigned seq = 0;
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {  // rep 0 warms up
            auto t0 = std::chrono::steady_clock::now();
            for (int c = 0; c < calls; ++c) {
                memcpy(hin, x.data(), n * 4);
                if (mode == 0) {
                    hipLaunchKernelGGL(work, dim3(1), dim3(256), 0, nullptr, din, dout, n);
                    CK(hipEventRecord(ev, nullptr));
                    CK(hipEventSynchronize(ev));
                } else if (mode == 1) {
                    hipLaunchKernelGGL(work, dim3(1), dim3(256), 0, nullptr, din, dout, n);
                    CK(hipStreamSynchronize(nullptr));
                } else if (mode == 2) {
                    ++seq;
                    hipLaunchKernelGGL(work, dim3(1), dim3(256), 0, nullptr, din, dout, n);
                    hipLaunchKernelGGL(flag, dim3(1), dim3(1), 0, nullptr, dflag, seq);
                    while (__atomic_load_n(hflag, __ATOMIC_ACQUIRE) != seq) {}
                } else {
                    ++seq;
                    hipLaunchKernelGGL(work_and_flag, dim3(1), dim3(256), 0, nullptr, din, dout, n, dflag, seq);
                    while (__atomic_load_n(hflag, __ATOMIC_ACQUIRE) != seq) {}
                }
                memcpy(y.data(), hout, n * 4);
                if (y[7] != 0.75f) {
                    printf("mode %d call %d: wrong result %f\n", mode, c, y[7]);
                    return 2;
                }
                hout[7] = 0.f;
            }
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / calls;
            if (rep) printf("mode %c: %.2f us per call\n", "ABCD"[mode], us);
        }
        CK(hipDeviceSynchronize());
    }
    return 0;
}
