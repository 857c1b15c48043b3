// Shared device/host helpers for librfuse_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/rfuse.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

void rf_set_error(const char* fmt, ...);

#define RF_REQUIRE(cond, code, ...)            \
    do {                                       \
        if (!(cond)) {                         \
            rf_set_error(__VA_ARGS__);         \
            return (code);                     \
        }                                      \
    } while (0)

#define RF_CHECK_LAUNCH(name)                                                    \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            rf_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return RF_E_LAUNCH;                                                  \
        }                                                                        \
    } while (0)

static inline bool rf_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int rf_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline int rf_round_up(int v, int m) { return (v + m - 1) / m * m; }

// wave64 sum reduction; result valid in every lane
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
