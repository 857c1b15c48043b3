// Error reporting + ABI version for librfuse_hip.so.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void rf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* rf_last_error(void) { return g_err; }
extern "C" int rf_abi_version(void) { return 1; }
