// capi_common.hpp - error plumbing shared by the translation units behind include/adsp.h
#pragma once
#include <hip/hip_runtime.h>

namespace adsp {
// records the message for adsp_last_error() (thread-local) and returns `code`
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace adsp

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return adsp::fail(ADSP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)
