// Per attention row: theta / phi features -> scores, switch, weights (reference model/attention.py:92-107).
// One definition for both attention routes (k_attn_fuse, k_attn_weights) so that they agree bit for bit: plain
// sequential fp32 sums over the f features, every product and sum rounded (files including this are built with
// -ffp-contract=off), F.normalize's v / max(||v||, 1e-12), torch's softmax (exp(z - zmax) / sum) and
// gumbel_softmax(hard=True) forward value (y_hard - y_soft) + y_soft.
#pragma once
#include "common.h"

#define RF_MAX_K 16

// scores of one row.  F > 0: compile-time feature width (a multiple of 4, rows 16-byte aligned): the row lives in
// registers and is read with 16-byte loads; F == 0: any width f, scalar loops.  Same operations in the same order.
template <int F>
__device__ __forceinline__ float rf_attn_row_scores(const float* __restrict__ xf_row, const float* __restrict__ pf_rows, int K, int f,
                                                    float (&sc)[RF_MAX_K]) {
    float smax = -INFINITY;
    if (F > 0) {
        float xv[F > 0 ? F : 1];
#pragma unroll
        for (int i = 0; i < F; i += 4) {
            const float4 t = *reinterpret_cast<const float4*>(xf_row + i);
            xv[i] = t.x; xv[i + 1] = t.y; xv[i + 2] = t.z; xv[i + 3] = t.w;
        }
        float n2 = 0.f;
#pragma unroll
        for (int i = 0; i < F; ++i) n2 += xv[i] * xv[i];
        const float xden = fmaxf(sqrtf(n2), 1e-12f);
#pragma unroll
        for (int i = 0; i < F; ++i) xv[i] = xv[i] / xden;
#pragma unroll
        for (int k = 0; k < RF_MAX_K; ++k) {
            sc[k] = -INFINITY;
            if (k < K) {
                float pv[F > 0 ? F : 1];
#pragma unroll
                for (int i = 0; i < F; i += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(pf_rows + (size_t)k * F + i);
                    pv[i] = t.x; pv[i + 1] = t.y; pv[i + 2] = t.z; pv[i + 3] = t.w;
                }
                float pn2 = 0.f;
#pragma unroll
                for (int i = 0; i < F; ++i) pn2 += pv[i] * pv[i];
                const float pden = fmaxf(sqrtf(pn2), 1e-12f);
                float dot = 0.f;
#pragma unroll
                for (int i = 0; i < F; ++i) dot += xv[i] * (pv[i] / pden);
                sc[k] = dot;
                smax = fmaxf(smax, dot);
            }
        }
    } else {
        float n2 = 0.f;
        for (int i = 0; i < f; ++i) n2 += xf_row[i] * xf_row[i];
        const float xden = fmaxf(sqrtf(n2), 1e-12f);
#pragma unroll
        for (int k = 0; k < RF_MAX_K; ++k) {
            sc[k] = -INFINITY;
            if (k < K) {
                const float* p = pf_rows + (size_t)k * f;
                float pn2 = 0.f;
                for (int i = 0; i < f; ++i) pn2 += p[i] * p[i];
                const float pden = fmaxf(sqrtf(pn2), 1e-12f);
                float dot = 0.f;
                for (int i = 0; i < f; ++i) dot += (xf_row[i] / xden) * (p[i] / pden);
                sc[k] = dot;
                smax = fmaxf(smax, dot);
            }
        }
    }
    return smax;
}

__device__ __forceinline__ void rf_attn_row_weights(const float* __restrict__ xf_row, const float* __restrict__ pf_rows,
                                                    const float* __restrict__ noise_row, int K, int f, int mode, float sharpness,
                                                    float (&sc)[RF_MAX_K], float (&w)[RF_MAX_K], float& sw) {
    const float smax = f == 32 ? rf_attn_row_scores<32>(xf_row, pf_rows, K, f, sc) : rf_attn_row_scores<0>(xf_row, pf_rows, K, f, sc);
    sw = fmaxf(smax, 0.f);                                    // relu(max_k scores), model/attention.py:99
    if (mode == RF_ATTN_SOFTMAX) {
        // softmax(sharpness * scores): z = sharpness*s rounded first, then exp(z - zmax) / sum as torch does
        const float zmax = sharpness * smax;
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < RF_MAX_K; ++k) {
            w[k] = k < K ? expf(sharpness * sc[k] - zmax) : 0.f;
            den += w[k];
        }
#pragma unroll
        for (int k = 0; k < RF_MAX_K; ++k) w[k] = w[k] / den;
    } else {
        // gumbel_softmax(logits = 25*scores, tau = 1, hard = True): y_hard - y_soft + y_soft
        float lg[RF_MAX_K], lmax = -INFINITY;
        int arg = 0;
#pragma unroll
        for (int k = 0; k < RF_MAX_K; ++k) {
            lg[k] = k < K ? sc[k] * 25.f + noise_row[k] : -INFINITY;
            if (lg[k] > lmax) { lmax = lg[k]; arg = k; }
        }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < RF_MAX_K; ++k) {
            w[k] = k < K ? expf(lg[k] - lmax) : 0.f;
            den += w[k];
        }
#pragma unroll
        for (int k = 0; k < RF_MAX_K; ++k) {
            const float ys = w[k] / den;
            const float yh = (k == arg) ? 1.f : 0.f;
            w[k] = (yh - ys) + ys;
        }
    }
}
