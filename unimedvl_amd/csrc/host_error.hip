// Error plumbing of the C ABI (host side): one thread-local message per library.  Compiled into libunimedvl_hip.so and - with
// UMV_EXPERIMENTAL_LIB - into libunimedvl_hip_experimental.so, which reports through umv_exp_last_error().
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";
void umv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
#ifdef UMV_EXPERIMENTAL_LIB
extern "C" const char* umv_exp_last_error(void) { return g_err; }
#else
extern "C" const char* umv_last_error(void) { return g_err; }
extern "C" int umv_version(void) { return 101; }
#endif
