// Dev probe (GPU box) for the "two-stream hazard" of DESIGN 4.7: synthetic aggressor kernels with a chosen VGPR footprint and instruction
// mix, to run beside small fp32-MFMA convs of the library on another stream (tools/hazard_probe.py compares the victims' bits).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/hazard_probe.so tools/micro/hazard_probe.hip
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// NACC accumulator tiles of 4 VGPRs stay live through the loop.  KIND 0: f16 MFMA 16x16x32; 1: fp32 MFMA 16x16x4; 2: VALU FMAs only
template <int NACC, int KIND, int WPE>
__global__ __launch_bounds__(256, WPE) void k_aggressor(float* out, int iters) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){(float)threadIdx.x, (float)i, 1.f, 2.f};
    h8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x & 15) + j); b[j] = (_Float16)(0.002f * j + 1.f); }
    const float af = 1.0001f, bf = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
            else if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[i], 0, 0, 0);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][r] = fmaf(acc[i][r], af, bf);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

#define LAUNCH(NACC, KIND, WPE) hipLaunchKernelGGL((k_aggressor<NACC, KIND, WPE>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters)
// regs: 0 -> 24 tiles (~110 VGPRs, 4 waves / SIMD), 1 -> 36 tiles (~160 VGPRs, 3), 2 -> 56 tiles (~240 VGPRs, 2)
extern "C" int launch_aggressor(int kind, int regs, int blocks, int iters, float* out, void* stream) {
    if (kind == 0 && regs == 0) LAUNCH(24, 0, 4);
    else if (kind == 0 && regs == 1) LAUNCH(36, 0, 3);
    else if (kind == 0 && regs == 2) LAUNCH(56, 0, 2);
    else if (kind == 1 && regs == 0) LAUNCH(24, 1, 4);
    else if (kind == 1 && regs == 1) LAUNCH(36, 1, 3);
    else if (kind == 1 && regs == 2) LAUNCH(56, 1, 2);
    else if (kind == 2 && regs == 0) LAUNCH(24, 2, 4);
    else if (kind == 2 && regs == 1) LAUNCH(36, 2, 3);
    else if (kind == 2 && regs == 2) LAUNCH(56, 2, 2);
    else return -1;
    return (int)hipGetLastError();
}

// Synthetic victim: every wave runs the same chain of K fp32 MFMAs (16x16x4) on the same operands; out[block][lane][4].  All blocks must
// return the same bits -- tools/hazard_probe.py compares them with each other and with two host emulations of the chain.
__global__ __launch_bounds__(64) void k_victim(const float* __restrict__ a, const float* __restrict__ b, int K, float* __restrict__ out) {
    // operands first, then the K (= 64) MFMAs back to back on one accumulator, as a conv kernel's k-loop issues them
    float av[64], bv[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) { av[k] = a[k * 64 + threadIdx.x]; bv[k] = b[k * 64 + threadIdx.x]; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 64; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k], bv[k], acc, 0, 0, 0);
    reinterpret_cast<f32x4*>(out)[blockIdx.x * 64 + threadIdx.x] = acc;
}

extern "C" int launch_victim(const float* a, const float* b, int K, int blocks, float* out, void* stream) {
    hipLaunchKernelGGL(k_victim, dim3(blocks), dim3(64), 0, (hipStream_t)stream, a, b, K, out);
    return (int)hipGetLastError();
}
