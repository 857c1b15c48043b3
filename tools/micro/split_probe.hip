// Dev probe (GPU box): fp32 GEMM arithmetic on the f16 matrix cores by operand splitting.
//   x = h + l / S exactly-ish:  h = f16(x), l = f16((x - h) * S), S = 2^11  (|x - h| <= 2^-11 |x|, so l is a full-precision f16 of the same
//   magnitude class as x; representation error <= 2^-22 |x|).  a*b ~ ah*bh + (ah*bl + al*bh) / S, the dropped al*bl term is 2^-22 relative.
//   Every f16 x f16 product is exact in fp32; the two sums are kept in separate fp32 accumulators (hi, lo) and combined once at the end.
// Part 1: error against float64 of (a) the fp32 MFMA chain v_mfma_f32_16x16x4_f32, (b) f16 split, 3 x v_mfma_f32_16x16x32_f16,
//         (c) bf16 three-piece split, 6 MFMAs, for conv-like operands (activations = relu(N(0,1)), weights = U(-1,1)/sqrt(K)).
// Part 2: what the split form sustains with its operands read from LDS (ds_read_b128), 16 output tiles per wave.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/split_probe tools/micro/split_probe.hip && /tmp/split_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
#define SPLIT_S 2048.0f

// one wave per 16 x 16 output tile; A [M][K] row-major, B [N][K] row-major (k contiguous for both)
__global__ void k_f32(const float* A, const float* B, float* D, int K, int N) {
    const int l = threadIdx.x, m0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(size_t)(m0 + (l & 15)) * K + k + (l >> 4)], B[(size_t)(n0 + (l & 15)) * K + k + (l >> 4)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(size_t)(m0 + 4 * (l >> 4) + r) * N + n0 + (l & 15)] = acc[r];
}
template <int MODE>      // 0: hi and lo accumulators; 1: everything into one accumulator (lo products unscaled: l = f16(x - h)); 2: hh only
__global__ void k_f16x3(const float* A, const float* B, float* D, int K, int N) {
    const int l = threadIdx.x, m0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
    f32x4 hi = {0.f, 0.f, 0.f, 0.f}, lo = {0.f, 0.f, 0.f, 0.f};
    const float S = MODE == 1 ? 1.0f : SPLIT_S;
    for (int k = 0; k < K; k += 32) {
        h8 ah, al, bh, bl;
        for (int j = 0; j < 8; ++j) {
            const int kk = k + 8 * (l >> 4) + j;
            const float a = kk < K ? A[(size_t)(m0 + (l & 15)) * K + kk] : 0.f, b = kk < K ? B[(size_t)(n0 + (l & 15)) * K + kk] : 0.f;
            ah[j] = (_Float16)a; al[j] = (_Float16)((a - (float)ah[j]) * S);
            bh[j] = (_Float16)b; bl[j] = (_Float16)((b - (float)bh[j]) * S);
        }
        hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, hi, 0, 0, 0);
        if (MODE == 0) { lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, lo, 0, 0, 0); lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, lo, 0, 0, 0); }
        if (MODE == 1) { hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, hi, 0, 0, 0); hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, hi, 0, 0, 0); }
    }
    for (int r = 0; r < 4; ++r) D[(size_t)(m0 + 4 * (l >> 4) + r) * N + n0 + (l & 15)] = hi[r] + lo[r] * (1.0f / SPLIT_S);
}
__global__ void k_bf16x6(const float* A, const float* B, float* D, int K, int N) {
    const int l = threadIdx.x, m0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
    f32x4 hi = {0.f, 0.f, 0.f, 0.f}, lo = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 32) {
        b8 a0, a1, a2, b0, b1, b2;
        for (int j = 0; j < 8; ++j) {
            const int kk = k + 8 * (l >> 4) + j;
            float a = kk < K ? A[(size_t)(m0 + (l & 15)) * K + kk] : 0.f, b = kk < K ? B[(size_t)(n0 + (l & 15)) * K + kk] : 0.f;
            a0[j] = (__bf16)a; a -= (float)a0[j]; a1[j] = (__bf16)a; a -= (float)a1[j]; a2[j] = (__bf16)a;
            b0[j] = (__bf16)b; b -= (float)b0[j]; b1[j] = (__bf16)b; b -= (float)b1[j]; b2[j] = (__bf16)b;
        }
        hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, hi, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b2, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b0, lo, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(size_t)(m0 + 4 * (l >> 4) + r) * N + n0 + (l & 15)] = hi[r] + lo[r];
}

// ---- rate: MB x NB tiles of 16 x 16 per wave, two accumulator sets, operands by ds_read_b128 ----
template <int MB, int NB, int WPS, int MODE>     // MODE 0: operands stay in registers; 1: re-read from LDS every k-step
__global__ __launch_bounds__(512, WPS / 2) void k_rate(float* out, int steps) {
    extern __shared__ h8 lds[];                   // 4096 entries of 16 B = 64 KB
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 512) { h8 v; for (int j = 0; j < 8; ++j) v[j] = (_Float16)(1e-3f * (float)((i + j) & 63)); lds[i] = v; }
    __syncthreads();
    f32x4 hi[MB][NB], lo[MB][NB];
    for (int m = 0; m < MB; ++m) for (int n = 0; n < NB; ++n) { hi[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; lo[m][n] = hi[m][n]; }
    h8 ah[MB], al[MB], bh[NB], bl[NB];
    for (int m = 0; m < MB; ++m) { ah[m] = lds[(tid + 64 * m) & 4095]; al[m] = lds[(tid + 64 * m + 1024) & 4095]; }
    for (int n = 0; n < NB; ++n) { bh[n] = lds[(tid + 64 * n + 2048) & 4095]; bl[n] = lds[(tid + 64 * n + 3072) & 4095]; }
    for (int st = 0; st < steps; ++st) {
        if (MODE == 1) {
            const int o = (st & 15) * 67;
#pragma unroll
            for (int m = 0; m < MB; ++m) { ah[m] = lds[(tid + 64 * m + o) & 4095]; al[m] = lds[(tid + 64 * m + 1024 + o) & 4095]; }
#pragma unroll
            for (int n = 0; n < NB; ++n) { bh[n] = lds[(tid + 64 * n + 2048 + o) & 4095]; bl[n] = lds[(tid + 64 * n + 3072 + o) & 4095]; }
        }
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                hi[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[n], hi[m][n], 0, 0, 0);
                lo[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[n], lo[m][n], 0, 0, 0);
                lo[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[n], lo[m][n], 0, 0, 0);
            }
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int m = 0; m < MB; ++m) for (int n = 0; n < NB; ++n) s += hi[m][n] + lo[m][n];
    out[blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
}

static double urand() { return (double)rand() / RAND_MAX; }
static double nrand() { return sqrt(-2.0 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand()); }

template <typename F> static void rate(const char* name, F launch, int mb, int nb, int steps, int grid) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); }
    float ms; hipEventElapsedTime(&ms, a, b);
    hipError_t e = hipGetLastError();
    const double fl = (double)grid * 8 * steps * mb * nb * 2.0 * 16 * 16 * 32;          // fp32-equivalent flop (one product per three MFMAs)
    printf("%-58s %8.3f ms  %7.1f TFLOP/s fp32-equivalent (%.2f x the fp32 MFMA peak), f16 pipe %.1f %% of 2516  %s\n", name, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3, 3 * fl / ms / 1e9 / 25.16, e == hipSuccess ? "" : hipGetErrorString(e));
}

int main() {
    const int M = 512, N = 64;
    for (int K : {216, 864, 1376, 2592, 5184}) {
        std::vector<float> A((size_t)M * K), B((size_t)N * K);
        srand(1234 + K);
        for (auto& v : A) { double x = nrand() * 0.8 + 0.3; v = (float)(x > 0 ? x : 0); }
        for (auto& v : B) v = (float)((2 * urand() - 1) / sqrt((double)K));
        std::vector<double> ref((size_t)M * N), mag((size_t)M * N);
        for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) { double s = 0, g = 0; for (int k = 0; k < K; ++k) { double p = (double)A[(size_t)m * K + k] * B[(size_t)n * K + k]; s += p; g += fabs(p); } ref[(size_t)m * N + n] = s; mag[(size_t)m * N + n] = g; }
        float *dA, *dB, *dD;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, (size_t)M * N * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> D((size_t)M * N);
        const char* names[5] = {"fp32 MFMA chain (16x16x4)", "f16 split, 3 MFMAs, hi/lo accumulators", "f16 split, 3 MFMAs, one accumulator", "f16 hh only (1 MFMA)", "bf16 3-piece split, 6 MFMAs"};
        printf("K = %d   (error / sum|a b|: max, rms;   error / rms|result|: rms)\n", K);
        double r2 = 0; for (double v : ref) r2 += v * v; r2 = sqrt(r2 / ref.size());
        for (int v = 0; v < 5; ++v) {
            dim3 g(M / 16, N / 16);
            if (v == 0) hipLaunchKernelGGL(k_f32, g, dim3(64), 0, 0, dA, dB, dD, K, N);
            else if (v == 1) hipLaunchKernelGGL(k_f16x3<0>, g, dim3(64), 0, 0, dA, dB, dD, K, N);
            else if (v == 2) hipLaunchKernelGGL(k_f16x3<1>, g, dim3(64), 0, 0, dA, dB, dD, K, N);
            else if (v == 3) hipLaunchKernelGGL(k_f16x3<2>, g, dim3(64), 0, 0, dA, dB, dD, K, N);
            else hipLaunchKernelGGL(k_bf16x6, g, dim3(64), 0, 0, dA, dB, dD, K, N);
            hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
            double mx = 0, s2 = 0, e2 = 0;
            for (size_t i = 0; i < D.size(); ++i) { double e = fabs((double)D[i] - ref[i]); double q = e / mag[i]; mx = fmax(mx, q); s2 += q * q; e2 += e * e; }
            printf("  %-42s max %.3e  rms %.3e   |  rms err / rms result %.3e\n", names[v], mx, sqrt(s2 / D.size()), sqrt(e2 / D.size()) / r2);
        }
        hipFree(dA); hipFree(dB); hipFree(dD);
    }
    float* out; hipMalloc(&out, (size_t)8192 * 512 * 4);
    const int grid = 2048;
#define RATE(MB, NB, WPS, MODE, STEPS, NAME)                                                                                   \
    hipFuncSetAttribute((const void*)k_rate<MB, NB, WPS, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);                \
    rate(NAME, [&] { hipLaunchKernelGGL((k_rate<MB, NB, WPS, MODE>), dim3(grid), dim3(512), 65536, 0, out, STEPS); }, MB, NB, STEPS, grid)
    RATE(4, 4, 2, 0, 256, "4x4 tiles, 2 waves/SIMD, operands in registers");
    RATE(4, 4, 2, 1, 256, "4x4 tiles, 2 waves/SIMD, 16 ds_read_b128 per 48 MFMAs");
    RATE(4, 2, 4, 0, 512, "4x2 tiles, 4 waves/SIMD, operands in registers");
    RATE(4, 2, 4, 1, 512, "4x2 tiles, 4 waves/SIMD, 12 ds_read_b128 per 24 MFMAs");
    RATE(2, 2, 4, 0, 1024, "2x2 tiles, 4 waves/SIMD, operands in registers");
    RATE(2, 2, 4, 1, 1024, "2x2 tiles, 4 waves/SIMD, 8 ds_read_b128 per 12 MFMAs");
    return 0;
}
