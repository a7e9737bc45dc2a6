// Error plumbing and version entry points of liburesnet_hip.so.
#include "ur_common.h"
#include <string.h>

namespace {
thread_local char g_err[512] = "";
}

void ur_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ur_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ur_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return UR_ERR_LAUNCH;
    }
    return UR_OK;
}

extern "C" {
int ur_version(void) { return 100; }
const char* ur_last_error(void) { return g_err; }
}
