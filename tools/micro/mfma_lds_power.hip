// Dev microbenchmark (GPU box): what the F16 matrix pipe sustains when its A operands come out of LDS at the rates the conv kernels use -- R ds_read_b128 per 12
// v_mfma_f32_16x16x32_f16 (the dominant launch: 2 per 12; the 16-cout z-column kernels: ~4-6 per 12; the box kernel of round 3: 8 per 12) -- random operands,
// 2 waves per SIMD, conflict-free reads.  Is the ~1.25 PFLOP/s every MFMA-heavy launch of the step plateaus at a property of LDS-fed MFMA streams
// under the power limit, or do the kernels lose the rest to their own stalls?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_lds tools/micro/mfma_lds_power.hip && /tmp/mfma_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int R>
__global__ __launch_bounds__(512, 2) void k(float* out, int steps) {
    __shared__ h8 img[4096];                                         // 64 KB
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    for (int i = threadIdx.x; i < 4096; i += 512) {
        h8 v;
        for (int j = 0; j < 8; ++j) { s = s * 1664525u + 1013904223u; v[j] = (_Float16)(((int)(s >> 16) - 32768) * (1.0f / 32768.f)); }
        img[i] = v;
    }
    __syncthreads();
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = img[(threadIdx.x + 64 * i) & 4095]; b[i] = img[(threadIdx.x + 64 * i + 1024) & 4095]; }
    f32x4 acc[4][4];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int st = 0; st < steps; ++st) {
        // 48 MFMAs per step; R reads per 12 MFMAs -> 4 R reads per step, spread over the A registers (each read lands in the register the NEXT step uses)
        h8 na[4] = {a[0], a[1], a[2], a[3]};
#pragma unroll
        for (int r = 0; r < 4 * R; ++r) {
            const h8 t = img[(wave * 512 + ((st * 16 + r) & 7) * 64 + lane) & 4095];
            na[r & 3] = (r < 4) ? t : na[r & 3] + t;                  // (more than 4 reads: folded together so that every read is used)
        }
#pragma unroll
        for (int rep = 0; rep < 3; ++rep)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[n], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = na[i];
    }
    float r = 0.f;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) r += acc[m][n][0] + acc[m][n][3];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
int main() {
    float* out;
    hipMalloc(&out, (size_t)4096 * 512 * 4);
    const int grid = 4096, steps = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 5; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, out, steps);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, out, steps);
            else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, out, steps);
            else if (mode == 3) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(512), 0, 0, out, steps);
            else hipLaunchKernelGGL(k<8>, dim3(grid), dim3(512), 0, 0, out, steps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 1 && ms < best) best = ms;
        }
        const int R[5] = {0, 1, 2, 4, 8};
        const double flops = (double)grid * 8 * steps * 48 * 16384.0;
        printf("%d ds_read_b128 per 12 MFMAs: %8.3f ms  %7.1f TFLOP/s  (%.3f of 2500)   LDS %.0f B/clk/CU at that rate\n", R[mode], best, flops / best / 1e9,
               flops / best / 1e9 / 2500.0, R[mode] ? (double)grid * 8 * steps * 4 * R[mode] * 1024.0 / (best * 1e-3) / 256 / 2.4e9 : 0.0);
    }
    return 0;
}
