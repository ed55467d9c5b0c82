// error.hip — thread-local error string + ABI version of libdaspeech_hip.so
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

namespace dsp {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("%s: %s", what, hipGetErrorString(e)); return (int)e; }
    return DSP_OK;
}
}  // namespace dsp

extern "C" int dsp_abi_version(void) { return DSP_ABI_VERSION; }
extern "C" const char* dsp_last_error(void) { return dsp::g_err; }
