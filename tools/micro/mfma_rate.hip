// Dev microbenchmark (GPU box): what v_mfma_f32_16x16x4_f32 sustains in the shape of the decoder kernel's tap loop -- 8 waves per
// workgroup, 2 workgroups per CU (4 waves per SIMD), 16 independent accumulator tiles per wave (MB 4 x NB 4) -- as pure MFMAs, with
// the tap loop's LDS operand reads (one s_waitcnt per k-step), and with a workgroup barrier every 27 k-steps.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/micro/mfma_rate.hip && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512, 4) void k(float* out, int steps) {
    __shared__ float lds[16384 + 2048];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += 512) lds[i] = 1e-3f * (float)(i & 255);
    if (MODE == 3)      // one LDS-DMA anywhere in the kernel: hipcc's s_waitcnt insertion then treats every later LDS wait as lgkmcnt(0)
        __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)(out + tid * 4), (__attribute__((address_space(3))) void*)(lds + 16384 + (tid >> 6) * 256), 16, 0, 0);
    __syncthreads();
    f32x4 acc[4][4];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float av[2][4], bv[2][4];
    for (int i = 0; i < 4; ++i) { av[0][i] = lds[tid + i * 512]; bv[0][i] = lds[8192 + tid + i * 512]; }
    for (int st = 0; st < steps; st += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 1) {
                const int o = ((st + h + 1) & 15) * 37;
                const int at = MODE == 4 ? ((tid & 15) * 24 + ((tid >> 4) & 3) * 1000 + (tid >> 6) * 2) : tid;    // MODE 4: the decoder kernel's lattice stride (4-way bank conflicts)
#pragma unroll
                for (int i = 0; i < 4; ++i) { av[h ^ 1][i] = lds[(at + i * 512 + o) & 8191]; bv[h ^ 1][i] = lds[8192 + ((tid + i * 512 + o) & 8191)]; }
                if (MODE == 5) {                                        // a 1-KiB LDS-DMA per wave and k-step, five in flight (the phase-B weight ring)
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)(out + ((st + h) & 63) * 4096 + (tid & 63) * 4),
                                                     (__attribute__((address_space(3))) void*)(lds + 16384 + (tid >> 6) * 256), 16, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) { av[h ^ 1][i] = av[h][i]; bv[h ^ 1][i] = bv[h][i]; }
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][m], bv[h][n], acc[m][n], 0, 0, 0);
            if (MODE >= 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            }
        }
        if (MODE == 2 && (st % 28) == 26) __syncthreads();
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) s += acc[m][n];
    out[blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
}
int main() {
    float* out;
    hipMalloc(&out, (size_t)8192 * 512 * 4);
    const int steps = 2 * 28 * 16, grid = 8192;                       // ~ the dominant launch: 8192 workgroups, 896 k-steps of 16 MFMAs
    const char* names[6] = {"pure MFMA (operands in registers)", "+ LDS operand reads, one behind every other MFMA", "+ a workgroup barrier every 28 k-steps", "LDS reads + one LDS-DMA in the kernel (conservative s_waitcnt)", "LDS reads with the lattice stride (bank conflicts)", "LDS reads + a 1-KiB LDS-DMA per wave and k-step"};
    for (int mode = 0; mode < 6; ++mode) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, out, steps);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, out, steps);
            else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, out, steps);
            else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(512), 0, 0, out, steps);
            else if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(512), 0, 0, out, steps);
            else hipLaunchKernelGGL(k<5>, dim3(grid), dim3(512), 0, 0, out, steps);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        const double flops = (double)grid * 8 * steps * 16 * 2048.0;
        printf("%-55s %8.3f ms  %6.1f TFLOP/s  (%.1f %% of 157.3)\n", names[mode], ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573);
    }
    return 0;
}
