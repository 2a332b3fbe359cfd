// Error plumbing of libunimedvl_hip_experimental.so (host side): one thread-local message, read through umv_exp_last_error().
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
extern "C" const char* umv_exp_last_error(void) { return g_err; }
