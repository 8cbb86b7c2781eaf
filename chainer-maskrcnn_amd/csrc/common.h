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
// Hand-written selection / sorting of unsigned 64-bit keys (sort.hip; no library): segments = images.
// select_kth: kth[seg * kth_stride] = the kreq[seg * kreq_stride]-th largest (descending) or smallest key of the segment (radix select,
// 11 bits per launch); top_k_sorted: the (up to) cap largest keys carrying valid_bit, in descending order (select + compact
// + bitonic sort in LDS; cap <= 16384).  Keys must be < 2^key_bits; workspaces from the *_ws_bytes queries.
size_t select_ws_bytes(int nseg, int key_bits);
int select_kth(const unsigned long long *keys, int nseg, size_t seg_len, int key_bits, bool descending, const unsigned *kreq,
               int kreq_stride, unsigned long long *kth, int kth_stride, void *ws, hipStream_t st, bool hist_is_zero = false);
size_t topk_ws_bytes(int nseg, int key_bits, int cap);
// words of a top_k_sorted workspace that must be initialised before its first kernel (topk_init_of); prepared = the caller's own previous
// kernel on the stream has done it
struct TopkInit { unsigned *zero; int nzero; unsigned *count; int ncount; unsigned *kreq; int nk; unsigned kval; };
TopkInit topk_init_of(int nseg, int key_bits, int cap, void *ws);
int top_k_sorted(const unsigned long long *keys, int nseg, size_t seg_len, int key_bits, unsigned long long valid_bit, int cap,
                 unsigned long long *out, size_t out_stride, void *ws, hipStream_t st, bool prepared = false);
}  // namespace mrcnn
