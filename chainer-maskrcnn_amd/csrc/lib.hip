// ABI version + thread-local error string of libmrcnn_hip.so.
#include "common.h"
#include <cstring>

namespace mrcnn {
static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace mrcnn

extern "C" int mrcnn_abi_version(void) { return MRCNN_ABI_VERSION; }
extern "C" const char *mrcnn_last_error(void) { return mrcnn::g_err; }
