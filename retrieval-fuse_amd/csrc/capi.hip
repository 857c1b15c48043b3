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

// Test harness: fill the LDS of every CU with NaN bit patterns.  LDS is not cleared between workgroups (or processes): a kernel that
// reads LDS it never wrote sees whatever ran there before -- usually harmless zeros in a fresh process, rarely something else.
// tests/conftest.py calls this before every GPU test so that such a read fails every time.
__global__ __launch_bounds__(1024) void k_poison_lds(unsigned* sink) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) lds[i] = 0x7fc0dead;
    __syncthreads();
    if (lds[(threadIdx.x * 37) % (160 * 1024 / 4)] == 12345u && sink) sink[0] = 1;      // keeps the stores alive
}

extern "C" int rf_debug_poison_lds(void* stream) {
    static RfLdsOptIn opt_in;
    if (int rc = opt_in.ensure(reinterpret_cast<const void*>(k_poison_lds), 160 * 1024, "rf_debug_poison_lds")) return rc;
    hipLaunchKernelGGL(k_poison_lds, dim3(2048), dim3(1024), 160 * 1024, (hipStream_t)stream, (unsigned*)nullptr);
    RF_CHECK_LAUNCH("rf_debug_poison_lds");
    return RF_OK;
}
