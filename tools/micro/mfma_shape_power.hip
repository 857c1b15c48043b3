// Dev microbenchmark (GPU box): what the F16 matrix pipe SUSTAINS on random operands (the clock follows power) for the two shapes of the dense f16 MFMA --
// v_mfma_f32_16x16x32_f16 (16 cycles, 8 operand VGPRs per 16 K flop) and v_mfma_f32_32x32x16_f16 (32 cycles, 8 operand VGPRs per 32 K flop: half the
// operand traffic per flop) -- operands in registers, 2 waves per SIMD, 16 independent accumulator tiles' worth of registers per wave, launches of ~2 ms.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape tools/micro/mfma_shape_power.hip && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__device__ inline h8 rnd(unsigned& s, bool zero) {
    h8 v;
    for (int j = 0; j < 8; ++j) { s = s * 1664525u + 1013904223u; v[j] = zero ? (_Float16)0.f : (_Float16)(((int)(s >> 16) - 32768) * (1.0f / 32768.f)); }
    return v;
}
template <int SHAPE, bool ZERO>
__global__ __launch_bounds__(512, 2) void k(float* out, int steps) {
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = rnd(s, ZERO); b[i] = rnd(s, ZERO); }
    float r = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[4][4];
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int st = 0; st < steps; ++st) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) r += acc[m][n][0] + acc[m][n][3];
    } else {
        f32x16 acc[2][2];
        for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
        for (int st = 0; st < steps; ++st) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * kk + m], b[2 * kk + n], acc[m][n], 0, 0, 0);
        }
        for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) r += acc[m][n][0] + acc[m][n][15];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
int main() {
    float* out;
    hipMalloc(&out, (size_t)4096 * 512 * 4);
    const int grid = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass)
    for (int mode = 0; mode < 4; ++mode) {
        const int steps = 6000;
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL((k<16, false>), dim3(grid), dim3(512), 0, 0, out, steps);
            else if (mode == 1) hipLaunchKernelGGL((k<32, false>), dim3(grid), dim3(512), 0, 0, out, steps);
            else if (mode == 2) hipLaunchKernelGGL((k<16, true>), dim3(grid), dim3(512), 0, 0, out, steps);
            else hipLaunchKernelGGL((k<32, true>), dim3(grid), dim3(512), 0, 0, out, steps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 1 && ms < best) best = ms;
        }
        // per wave and step: 16 MFMAs of 16x16x32 (16384 flop each) = 8 MFMAs of 32x32x16 (32768 flop each) = 262144 flop
        const double flops = (double)grid * 8 * steps * 262144.0;
        const char* names[4] = {"16x16x32 random operands", "32x32x16 random operands", "16x16x32 zero operands", "32x32x16 zero operands"};
        printf("%-28s %8.3f ms  %7.1f TFLOP/s  (%.3f of 2500)\n", names[mode], best, flops / best / 1e9, flops / best / 1e9 / 2500.0);
    }
    return 0;
}
