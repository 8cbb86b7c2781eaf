// 64-bit key radix sort used by the proposal path (top-12000 by score) and by the anchor / proposal
// samplers (k smallest random keys).  Composite keys are unique, so the result is deterministic.
#include "common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace mrcnn {
int sort_u64(const unsigned long long *in, unsigned long long *out, size_t n, bool descending, void *tmp,
             size_t *tmp_bytes, hipStream_t st, unsigned end_bit) {
    hipError_t e;          // keys only differ in bits [0, end_bit): fewer radix passes for narrow composite keys
    if (descending) e = rocprim::radix_sort_keys_desc(tmp, *tmp_bytes, in, out, n, 0, end_bit, st);
    else e = rocprim::radix_sort_keys(tmp, *tmp_bytes, in, out, n, 0, end_bit, st);
    if (e != hipSuccess) {
        set_error("rocprim radix sort failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}
}  // namespace mrcnn
