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

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: a process that drives several GPUs has to
// opt in on each of them.  One instance per kernel instantiation (function-local static); bit d = "done on device d".
#include <atomic>
struct RfLdsOptIn {
    std::atomic<unsigned long long> done[4];
    RfLdsOptIn() { for (auto& d : done) d.store(0ull); }
    int ensure(const void* kernel, int bytes, const char* who) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) { rf_set_error("%s: cannot query the current device", who); return RF_E_LAUNCH; }
        const unsigned long long bit = 1ull << (dev & 63);
        if (done[dev >> 6].load(std::memory_order_acquire) & bit) return RF_OK;
        hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) { rf_set_error("%s: cannot raise the LDS limit to %d bytes: %s", who, bytes, hipGetErrorString(e)); return RF_E_LAUNCH; }
        done[dev >> 6].fetch_or(bit, std::memory_order_release);
        return RF_OK;
    }
};

static inline bool rf_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int rf_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline int rf_round_up(int v, int m) { return (v + m - 1) / m * m; }

// Workgroups of a PERSISTENT launch (k_conv3_split_zc / _zcm, k_conv3_up_split_boxp: 512 threads, <= 80 KB of LDS, <= 128 VGPRs -> two resident per CU): two per
// CU of the current device (hipDeviceProp_t::multiProcessorCount, 256 on MI355X -> 512; ADVICE r4: this was a literal 512 in three launchers and in the
// scratch-slot count), times RF_PERSIST_ROUNDS.  One round = every workgroup resident for the whole launch, a static share of the items each -- and then
// the launch ends when its UNLUCKIEST workgroup does: inside the pipelined step the other streams' kernels (the chunk-level U-Net's ~40 small launches, the
// top-k scan) hold a CU for 10-100 us at a time, the workgroup that waits for that CU starts late and still has its full share in front of it (round 5:
// the U-Net backbone beside the back end cost the step 0.35 ms for ~0.1 ms of work, tools/overlap_parts.py).  With FOUR rounds the workgroups of a late CU's
// later rounds go to whichever CU is free: same-box A/B of the C2 step 1 / 2 / 4 / 8 rounds = 6.47 / 6.43 / 6.39 / 6.42 ms (tools/abn_bench.sh; 8 pays more
// prologues -- weights to LDS, first halo image -- than it balances).  The launchers and rf_conv3d_split_pre_pool_presplit_scratch_floats (a scratch
// slot per workgroup, slot = blockIdx.x) all take the count from here.
#ifndef RF_PERSIST_ROUNDS
#define RF_PERSIST_ROUNDS 4
#endif
static inline int rf_resident_wgs() {
    // per device of the process (a count cached at first call would be the first device's for every later one), cached: hipGetDeviceProperties is slow
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 512;
    if (dev < 64) {
        const int c = cached[dev].load(std::memory_order_relaxed);
        if (c > 0) return c;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) return 512;
    if (dev < 64) cached[dev].store(2 * prop.multiProcessorCount, std::memory_order_relaxed);
    return 2 * prop.multiProcessorCount;
}
static inline int rf_persistent_wgs() { return rf_resident_wgs() * RF_PERSIST_ROUNDS; }

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

// GroupNorm as the conv kernels apply it: y = (x - center) * scale + shift with, per (sample, channel),
//   center = fl32(mean),  scale = fl32(gamma * rstd),  shift = fl32(beta - (mean - center) * gamma * rstd).
// Subtracting the (fp32-representable) centre first keeps the product at the magnitude of the RESULT: the plain two-term form
// x*scale' + shift' carries an absolute error of eps * |mean * rstd * gamma|, which for near-constant inputs (a truncation-
// saturated TSDF patch: rstd up to 316) is 100x the fp32 resolution of y and was the largest single error source of the path
// (tools/error_budget.py, first layer of the retrieval backbone).  What fl32(mean) loses is folded into `shift` in float64.
__device__ __forceinline__ float4 gn_affine(double mean, double rstd, float gamma, float beta) {
    const double sc = (double)gamma * rstd;
    const float center = (float)mean;
    return make_float4(center, (float)sc, (float)((double)beta - (mean - (double)center) * sc), 0.f);
}
