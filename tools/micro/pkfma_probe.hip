// Dev probe (GPU box): do packed-fp32 VALU instructions return the same bits whatever else runs on their SIMD?  (the fp32 box conv's
// GroupNorm apply, compiled to v_pk_add_f32 / v_pk_fma_f32, moved by one ulp beside another wave's F16 MFMAs; scalar v_fma_f32 did not.)
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/_haz/pkfma.so tools/micro/pkfma_probe.hip       (driver: tools/pkfma_probe.py)
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// KIND 0: v_pk_fma_f32 d, a, b, c              1: v_pk_fma_f32 d, a, s, s op_sel (scale / shift broadcast from one register pair, as hipcc emits it)
//      2: v_pk_mul_f32                          3: v_pk_add_f32                  4: two scalar v_fma_f32 (control)
//      5: KIND 0 behind four fp32 MFMAs of the same wave (the conv kernel's situation)
template <int KIND>
__device__ __forceinline__ f32x2 op(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    if (KIND == 0 || KIND == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    else if (KIND == 1) { f32x2 s = {b.x, c.x}; asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(s)); }
    else if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if (KIND == 3) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(c));
    else if (KIND == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));                    // src1: low half broadcast
    else if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));     // src2: high half broadcast
    else if (KIND == 8) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));     // both, different registers
    else if (KIND == 9) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));                    // src0: low half broadcast
    else if (KIND == 10) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    else if (KIND == 11) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(c));
    else if (KIND == 12) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));    // src1: high half broadcast
    else if (KIND == 13) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));                   // src2: low half broadcast
    else if (KIND == 14) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));    // src0: halves swapped
    else if (KIND == 15) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));                      // src0: high half broadcast (k_convv_valu)
    else if (KIND == 16) {                                                                                                              // ... with an SGPR-pair src1
        const unsigned long long sb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(b.y)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(b.x));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(d) : "v"(a), "s"(sb), "v"(c));
    }
    else if (KIND == 17) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(c));                  // src1 halves swapped (k_l2_topk)
    else if (KIND == 18) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));  // k_conv3_up's commit
    else if (KIND == 19) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    else if (KIND == 20) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(c));                  // src0 halves swapped
    else { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d.x) : "v"(a.x), "v"(b.x), "v"(c.x)); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d.y) : "v"(a.y), "v"(b.y), "v"(c.y)); }
    return d;
}

constexpr int NV = 8;          // operand triples per lane
// in: [3][NV][64 lanes] f32x2 (the same for every wave); res: [blocks*4 waves][NV][64] f32x2 results of the last repetition;
// expect (optional): [NV][64] f32x2 -> mism[wave][NV][64] counts the repetitions whose result differed, bad[...] keeps one differing value
template <int KIND>
__global__ __launch_bounds__(256) void k_pk(const f32x2* __restrict__ in, f32x2* __restrict__ res, const f32x2* __restrict__ expect,
                                            int* __restrict__ mism, f32x2* __restrict__ bad, int reps) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x2 a[NV], b[NV], c[NV], e[NV], r[NV];
    int mm[NV];
    f32x2 bd[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        a[k] = in[(0 * NV + k) * 64 + lane]; b[k] = in[(1 * NV + k) * 64 + lane]; c[k] = in[(2 * NV + k) * 64 + lane];
        e[k] = expect ? expect[k * 64 + lane] : (f32x2){0.f, 0.f};
        mm[k] = 0; bd[k] = (f32x2){0.f, 0.f}; r[k] = (f32x2){0.f, 0.f};
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < reps; ++it) {
        if (KIND == 5) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b[j].x, acc, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            r[k] = op<KIND>(a[k], b[k], c[k]);
            if (expect && (__float_as_uint(r[k].x) != __float_as_uint(e[k].x) || __float_as_uint(r[k].y) != __float_as_uint(e[k].y))) { ++mm[k]; bd[k] = r[k]; }
        }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        res[((size_t)wave * NV + k) * 64 + lane] = r[k];
        if (expect) { mism[((size_t)wave * NV + k) * 64 + lane] = mm[k]; bad[((size_t)wave * NV + k) * 64 + lane] = bd[k]; }
    }
    if (KIND == 5 && acc[0] == 12345.678f) res[0] = (f32x2){acc[1], acc[2]};
}

extern "C" int launch_pk(int kind, int blocks, const void* in, void* res, const void* expect, void* mism, void* bad, int reps, void* stream) {
#define L(K) hipLaunchKernelGGL((k_pk<K>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f32x2*)in, (f32x2*)res, (const f32x2*)expect, (int*)mism, (f32x2*)bad, reps)
    switch (kind) { case 0: L(0); break; case 1: L(1); break; case 2: L(2); break; case 3: L(3); break; case 4: L(4); break; case 5: L(5); break; case 6: L(6); break; case 7: L(7); break; case 8: L(8); break; case 9: L(9); break; case 10: L(10); break; case 11: L(11); break; case 12: L(12); break; case 13: L(13); break; case 14: L(14); break; case 15: L(15); break; case 16: L(16); break; case 17: L(17); break; case 18: L(18); break; case 19: L(19); break; case 20: L(20); break; default: return -1; }
    return (int)hipGetLastError();
}
