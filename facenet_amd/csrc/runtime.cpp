// Error plumbing for the C ABI (thread-local message, no global mutable state otherwise).
#include "common.h"
#include "../../include/facenet_hip.h"
#include <cstdarg>
#include <cstdio>

namespace fn {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return FN_ELAUNCH;
    }
    return FN_OK;
}
}  // namespace fn

extern "C" const char* fn_last_error(void) { return fn::g_err; }
extern "C" int fn_abi_version(void) { return 1; }
