// error.hip — thread-local error string + ABI version of libdaspeech_hip.so
#include <mutex>
#include <vector>
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

namespace dsp {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a driver call on the host's launch path (r06: every launcher made it on EVERY launch — the
// acoustic stage issues ~450 kernels per batch, the DAG ops at small shapes are host-paced).  Once per (kernel, device) and size class instead:
// the attribute is only ever raised.
void set_max_dynamic_lds(const void* fn, int bytes) {
    struct Ent { const void* fn; int dev; int bytes; };
    static std::mutex mu;
    static std::vector<Ent> tab;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    for (auto& e : tab) {
        if (e.fn == fn && e.dev == dev) {
            if (e.bytes >= bytes) return;
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess) e.bytes = bytes;
            return;
        }
    }
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess) tab.push_back({fn, dev, bytes});
}
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("%s: %s", what, hipGetErrorString(e)); return (int)e; }
    return DSP_OK;
}
}  // namespace dsp

extern "C" int dsp_abi_version(void) { return DSP_ABI_VERSION; }
extern "C" const char* dsp_last_error(void) { return dsp::g_err; }
