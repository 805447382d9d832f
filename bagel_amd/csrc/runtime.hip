// Error convention + version for libbagel_hip.so.
// Every exported op returns 0 or a negative code; the message is thread-local and valid until the next
// call on the same thread.  Ops never allocate, never synchronise and hold no global mutable state.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

int bagel_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int bagel_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bagel_set_error(BAGEL_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return BAGEL_OK;
}

extern "C" int bagel_hip_version(void) { return 100; }
extern "C" const char* bagel_hip_last_error(void) { return g_err; }
extern "C" const char* bagel_hip_arch(void) { return "gfx950"; }
