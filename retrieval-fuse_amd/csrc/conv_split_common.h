// Shared by the split-operand box kernels (conv3d_split.hip, conv3d_split_zc.hip): the halo-box image in LDS, the operand split and the MFMA triple.
#pragma once
#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

namespace {
// halo box [10][10][10] in 16-byte slots, Y-MAJOR with a padded y stride: slot(z, y, x) = y * 104 + z * 10 + x.  An A operand is a ds_read_b128 of
// an m-block = 8 x by 2 y voxels, served in four fixed 16-lane groups (MI355X_MICROARCH.md, LDS): lanes {0-3, 12-15} of one tap and {4-11} of the
// next, i.e. x 0-3 of row y, x 4-7 of row y + 1 and the other halves one tap on.  With slot(z, y, x) = z * 100 + y * 10 + x (rounds 2-3) row y + 1
// sat 10 slots on: x = 6, 7 of it on the slots (mod 16) of x = 0, 1 of row y -- two LDS cycles per group for every A operand of every box kernel
// (SQ_LDS_BANK_CONFLICT: 0.36-0.46 of the active LDS cycles).  With the rows of a tile 104 = 8 (mod 16) slots apart the group covers 16 different
// slots except where two taps meet: 1.29 cycles per group over the 7 k-steps (model: tools/lds_bank_model.py).
constexpr int CS_SY = 104, CS_SZ = 10, CS_VOX = 1000, CS_SLOTS = 1040;
constexpr int CS_PLANE = CS_SLOTS * 16;                          // bytes of one (h or l) plane
constexpr int CS_BUF = 2 * CS_PLANE;
constexpr int CS_LDS_BYTES = 2 * CS_BUF;                         // 66,560
constexpr float CS_ACT_SCALE = 1.0f / 16, CS_W_SCALE = 16.0f, CS_LO = 2048.0f;
constexpr bool CS_S4_WIDE = true;
}   // namespace


__device__ __forceinline__ void cs_split8(const float (&y)[8], h8& h, h8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = __builtin_amdgcn_fmed3f(y[j] * CS_ACT_SCALE, -65504.f, 65504.f);
        const _Float16 hh = (_Float16)v;
        h[j] = hh;
        l[j] = (_Float16)fmaf(-CS_LO, (float)hh, v * CS_LO);          // (v - h) * 2^11: exact either way, one v_fma_mix instead of cvt + sub + mul
    }
}

template <int NB>
__device__ __forceinline__ void cs_mfma_block(f32x4 (&hi)[NB], f32x4 (&lo)[NB], const h8& ah, const h8& al, const h8 (&bh)[NB], const h8 (&bl)[NB]) {
#pragma unroll
    for (int n = 0; n < NB; ++n) hi[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[n], hi[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NB; ++n) lo[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[n], lo[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NB; ++n) lo[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[n], lo[n], 0, 0, 0);
}

// pre-split OUTPUT (whole 8^3 samples, 16 couts in one workgroup): the NEXT layer's GroupNorm -- its gamma / beta / groups / eps over this layer's couts --
// is applied in the epilogue from the sample's own statistics and the result written as that layer's pre-split input; null: off
struct SplitPreOut {
    h8* out;
    const float* gamma;
    const float* beta;
    int groups;
    float eps;
    // ... or the final decoder's pointwise head (reference model/refinement.py:48-61: Conv3d(nf, 1, 1) + bias -> tanh -> network_pred_to_df) applied to the
    // ReLU'd output in the epilogue: pw_out [n][1][edge^3] = (tanh(sum_c w[c] y[c] + b) + post_add) * post_mul, the nf-channel tensor is never written
    float* pw_out;
    const float* pw_w;
    const float* pw_b;
    float post_add, post_mul;
};
