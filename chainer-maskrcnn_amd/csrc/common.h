// Shared host-side helpers of libmrcnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/mrcnn_hip.h"

namespace mrcnn {

void set_error(const char *fmt, ...);

inline int fail_arg(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return code;
}

// Launch-error check: returns the hipError_t as a positive int.
#define MRCNN_HIP_TRY(expr)                                                          \
    do {                                                                             \
        hipError_t e__ = (expr);                                                     \
        if (e__ != hipSuccess) {                                                     \
            mrcnn::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                             __FILE__, __LINE__);                                    \
            return (int)e__;                                                         \
        }                                                                            \
    } while (0)

#define MRCNN_LAUNCH_CHECK() MRCNN_HIP_TRY(hipGetLastError())

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace mrcnn

constexpr int kWave = 64;  // gfx950 wavefront

namespace mrcnn {
// Device radix sort of 64-bit keys (sort.hip; rocPRIM device primitive, the only library call in
// the library).  tmp == nullptr: returns the temporary-storage size in *tmp_bytes.
int sort_u64(const unsigned long long *in, unsigned long long *out, size_t n, bool descending, void *tmp,
             size_t *tmp_bytes, hipStream_t st, unsigned end_bit = 64);
}  // namespace mrcnn
