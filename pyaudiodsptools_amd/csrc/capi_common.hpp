// capi_common.hpp - error plumbing shared by the translation units behind include/adsp.h
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <mutex>

namespace adsp {
// records the message for adsp_last_error() (thread-local) and returns `code`
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Small host calls of the standalone effects, the delay lines and the scans - one chunk of one channel per call is the reference's own
// call pattern (ModuleTests.py:95-214): ONE pair of pinned, mapped host buffers per device, which the kernel reads and writes over PCIe
// itself.  No hipMalloc / staging copies / hipFree per call (46 -> 2x us per call of 512 samples; the filters' apply_host_direct
// has worked this way since round 2).  The mutex is held from the copy into the window until the copy out of it.
struct HostWindow {
    std::mutex mu;
    char *in = nullptr, *out = nullptr;  // host addresses
    void *d_in = nullptr, *d_out = nullptr;  // the same memory as the device sees it
    size_t cap_in = 0, cap_out = 0;
    hipEvent_t done = nullptr;
};
constexpr size_t kHostWindowMax = size_t(1) << 20;  // larger calls keep their staging copies (DMA engines beat PCIe loads there)
HostWindow* host_window(int device_id);             // nullptr (and the error set) when device_id is out of range
int host_window_reserve(HostWindow& w, size_t in_bytes, size_t out_bytes);  // w.mu held, the device current
int host_window_wait(HostWindow& w, hipStream_t stream);                    // everything launched on `stream` so far is done
}  // namespace adsp

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return adsp::fail(ADSP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)
