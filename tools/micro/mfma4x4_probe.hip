// Dev probe (GPU box): operand / result layout of v_mfma_f32_4x4x1_16b_f32 and of v_permlane{16,32}_swap on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma4x4_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(float* d, int la, int lb) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(l == la ? 1.f : 0.f, l == lb ? 1.f : 0.f, acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[i * 64 + l] = acc[i];
}
__global__ void k_swap(unsigned* o) {
    const unsigned l = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(l, 100u + l, false, false);
    o[l] = r[0]; o[64 + l] = r[1];
    auto q = __builtin_amdgcn_permlane16_swap(l, 100u + l, false, false);
    o[128 + l] = q[0]; o[192 + l] = q[1];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4);
    float h[256];
    int bad = 0;
    for (int la = 0; la < 64; ++la)
        for (int j = 0; j < 4; ++j) {
            const int lb = (la / 4) * 4 + j;
            hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, d, la, lb);
            hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            int n = 0, reg = -1, lane = -1;
            for (int q = 0; q < 256; ++q) if (h[q] != 0.f) { ++n; reg = q / 64; lane = q % 64; }
            const int want_reg = la % 4, want_lane = (la / 4) * 4 + j;
            if (n != 1 || reg != want_reg || lane != want_lane) { if (bad++ < 8) printf("la %d lb %d -> n %d reg %d lane %d (expected reg %d lane %d)\n", la, lb, n, reg, lane, want_reg, want_lane); }
        }
    // cross-block must be zero
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, d, 0, 4);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int nz = 0; for (int q = 0; q < 256; ++q) nz += h[q] != 0.f;
    printf("4x4x1_16b: D[reg i][lane 4b+j] = A[lane 4b+i] * B[lane 4b+j]: %s (%d mismatches), cross-block nonzeros %d\n", bad ? "NO" : "yes", bad, nz);
    unsigned* o; hipMalloc(&o, 256 * 4);
    unsigned ho[256];
    hipLaunchKernelGGL(k_swap, dim3(1), dim3(64), 0, 0, o);
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    printf("permlane32_swap(a=l, b=100+l): a' ="); for (int l = 0; l < 64; l += 8) printf(" %u", ho[l]); printf(" | b' ="); for (int l = 0; l < 64; l += 8) printf(" %u", ho[64 + l]); printf("\n");
    printf("permlane16_swap(a=l, b=100+l): a' ="); for (int l = 0; l < 64; l += 8) printf(" %u", ho[128 + l]); printf(" | b' ="); for (int l = 0; l < 64; l += 8) printf(" %u", ho[192 + l]); printf("\n");
    return 0;
}
