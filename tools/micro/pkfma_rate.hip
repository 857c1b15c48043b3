// Dev microbenchmark (GPU box): issue rate of v_pk_fma_f32 with VGPR-pair vs SGPR-pair multiplicands, and of v_fma_f32.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pkfma tools/micro/pkfma_rate.hip && /tmp/pkfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ w, float* out, int iters) {
    v2 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (v2){0.f, (float)threadIdx.x};
    v2 x = {(float)threadIdx.x * 1e-3f, 1.0f};
    for (int it = 0; it < iters; ++it) {
        const float* wp = w + (it & 15) * 32;                 // wave-uniform -> scalar loads
        if (MODE == 0) {                                       // SGPR-pair multiplicand
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) { const v2 wv = {wp[2 * i], wp[2 * i + 1]}; acc[i] = __builtin_elementwise_fma(x, wv, acc[i]); }
        } else if (MODE == 1) {                                // VGPR-pair multiplicand
            v2 wv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) wv[i] = (v2){wp[2 * i] + x[0], wp[2 * i + 1]};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(x, wv[i], acc[i]);
        } else {                                               // scalar v_fma_f32, SGPR multiplicand
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) { acc[i][0] = fmaf(x[0], wp[2 * i], acc[i][0]); acc[i][1] = fmaf(x[1], wp[2 * i + 1], acc[i][1]); }
        }
    }
    v2 s = {0.f, 0.f};
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1];
}
int main() {
    float *w, *out;
    hipMalloc(&w, 4096); hipMemset(w, 0, 4096); hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 4096, grid = 4096;
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, w, out, iters);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, w, out, iters);
            else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, w, out, iters);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        const double flops = (double)grid * 256 * iters * 64 * 4;     // 64 pk_fma (or 128 fma) per iteration, 4 flop per pk_fma
        printf("mode %d (%s): %.3f ms  %.1f TFLOP/s\n", mode, mode == 0 ? "v_pk_fma_f32, SGPR-pair weights" : mode == 1 ? "v_pk_fma_f32, VGPR-pair weights" : "v_fma_f32, SGPR weights", ms, flops / ms / 1e9);
    }
    return 0;
}
