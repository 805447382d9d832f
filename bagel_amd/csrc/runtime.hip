// Error convention + version for libbagel_hip.so.
// Every exported op returns 0 or a negative code; the message is thread-local and valid until the next
// call on the same thread.  Ops never allocate, never synchronise and hold no global mutable state.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <mutex>

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

int bagel_enable_lds(const void* func, int bytes, const char* what) {
    constexpr int MAXF = 256, MAXDEV = 16;
    static std::mutex mu;
    static const void* funcs[MAXF];
    static int enabled[MAXF][MAXDEV];      // largest size enabled for (kernel, device); 0 = never
    static int nfuncs = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return bagel_set_error(BAGEL_ERR_LAUNCH, "%s: no current device", what);
    std::lock_guard<std::mutex> lock(mu);
    int i = 0;
    while (i < nfuncs && funcs[i] != func) ++i;
    if (i == nfuncs) {
        if (nfuncs == MAXF) return bagel_set_error(BAGEL_ERR_LAUNCH, "%s: LDS attribute table full", what);
        funcs[nfuncs++] = func;
    }
    if (enabled[i][dev] >= bytes) return BAGEL_OK;
    hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess)
        return bagel_set_error(BAGEL_ERR_LAUNCH, "%s: cannot enable %d bytes of LDS on device %d: %s", what, bytes, dev, hipGetErrorString(e));
    enabled[i][dev] = bytes;
    return BAGEL_OK;
}

extern "C" int bagel_hip_version(void) { return 100; }
extern "C" const char* bagel_hip_last_error(void) { return g_err; }
extern "C" const char* bagel_hip_arch(void) { return "gfx950"; }

// ---- hipGraph capture of a launch sequence (the per-token decode step) ----------------------------------------------
// The ops above only launch kernels on the caller's stream, so a step recorded once between begin/end replays with one
// host call per token.  Capture is "relaxed": the host framework may touch its allocator on other threads meanwhile.
extern "C" int bagel_graph_begin(hipStream_t stream) {
    BAGEL_REQUIRE(stream != nullptr, "graph_begin: the legacy default stream cannot be captured; use a side stream");
    hipError_t e = hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) return bagel_set_error(BAGEL_ERR_LAUNCH, "graph_begin: %s", hipGetErrorString(e));
    return BAGEL_OK;
}

extern "C" int bagel_graph_end(hipStream_t stream, void** exec_out) {
    BAGEL_REQUIRE(exec_out != nullptr, "graph_end: null output");
    *exec_out = nullptr;
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(stream, &graph);
    if (e != hipSuccess || graph == nullptr) {
        (void)hipGetLastError();
        return bagel_set_error(BAGEL_ERR_LAUNCH, "graph_end: capture failed: %s", hipGetErrorString(e));
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return bagel_set_error(BAGEL_ERR_LAUNCH, "graph_end: instantiate failed: %s", hipGetErrorString(e));
    *exec_out = (void*)exec;
    return BAGEL_OK;
}

extern "C" int bagel_graph_launch(void* exec, hipStream_t stream) {
    BAGEL_REQUIRE(exec != nullptr, "graph_launch: null graph");
    hipError_t e = hipGraphLaunch((hipGraphExec_t)exec, stream);
    if (e != hipSuccess) return bagel_set_error(BAGEL_ERR_LAUNCH, "graph_launch: %s", hipGetErrorString(e));
    return BAGEL_OK;
}

extern "C" int bagel_graph_destroy(void* exec) {
    if (exec) (void)hipGraphExecDestroy((hipGraphExec_t)exec);
    return BAGEL_OK;
}
