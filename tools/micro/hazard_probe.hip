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
