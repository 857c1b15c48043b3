// Patch attention for gfx950: Unfold3D / Fold3D as index remaps, the retrieved-feature regroup, and the fused
// normalise -> scores -> switch -> softmax | Gumbel-hard -> weighted sum of retrieved patches -> blend.
//
// Reference arithmetic being replaced: model/attention.py:84-113 (AttentionBlock.forward), :141-157
// (PatchedAttentionBlock.forward), :170-176 (Fold3D), :186-188 (Unfold3D).
// The per-row contractions are K dot products of length 32 and a K-term weighted sum of 128 values (K = 4..8):
// far too small and block-diagonal for MFMA; they run as wave-shuffle reductions, one wave per attention row, and
// the kernel is HBM-bound (reads (1+K)*(d+f) floats, writes d floats per row).
#include "common.h"
#include "attn_row.h"

// torch evaluates these expressions op by op (every product and sum rounded).  This file is built with
// -ffp-contract=off (csrc/build.py): hipcc's default -ffp-contract=fast fuses a*b+c in the backend, where neither
// __fmul_rn/__fadd_rn nor `#pragma clang fp contract(off)` reach.  Explicit fmaf() stays an FMA.

// rows[((b*r+p0)*r+p1)*r+p2][c][e0][e1][e2] = x[b][c][p0*e+e0][p1*e+e1][p2*e+e2]
__global__ __launch_bounds__(256) void k_unfold3d(const float* __restrict__ x, int b, int c, int s, int e, float* __restrict__ rows) {
    const int r = s / e;
    const size_t e3 = (size_t)e * e * e, total = (size_t)b * c * s * s * s;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        // i indexes the OUTPUT (rows) so writes are coalesced
        const int e2 = (int)(i % e), e1 = (int)((i / e) % e), e0 = (int)((i / ((size_t)e * e)) % e);
        const int cc = (int)((i / e3) % c);
        const size_t row = i / (e3 * c);
        const int p2 = (int)(row % r), p1 = (int)((row / r) % r), p0 = (int)((row / ((size_t)r * r)) % r);
        const size_t bb = row / ((size_t)r * r * r);
        rows[i] = x[(((bb * c + cc) * s + (p0 * e + e0)) * s + (p1 * e + e1)) * s + (p2 * e + e2)];
    }
}

__global__ __launch_bounds__(256) void k_fold3d(const float* __restrict__ rows, int b, int c, int s, int e, float* __restrict__ x) {
    const int r = s / e;
    const size_t e3 = (size_t)e * e * e, total = (size_t)b * c * s * s * s;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        // i indexes the OUTPUT volume
        const int d2 = (int)(i % s), d1 = (int)((i / s) % s), d0 = (int)((i / ((size_t)s * s)) % s);
        const int cc = (int)((i / ((size_t)s * s * s)) % c);
        const size_t bb = i / ((size_t)s * s * s * c);
        const size_t row = ((bb * r + d0 / e) * r + d1 / e) * r + d2 / e;
        x[i] = rows[(row * c + cc) * e3 + ((size_t)(d0 % e) * e + (d1 % e)) * e + (d2 % e)];
    }
}

static unsigned grid_for(size_t total) {
    const size_t want = (total + 255) / 256;
    return (unsigned)(want < 8192 ? (want ? want : 1) : 8192);
}

static int fold_check(const char* who, const void* a, const void* b2, int b, int c, int s, int e) {
    RF_REQUIRE(a && b2 && b > 0 && c > 0 && s > 0 && e > 0 && s % e == 0, RF_E_INVALID, "%s: bad arguments (s=%d e=%d)", who, s, e);
    return RF_OK;
}

extern "C" int rf_unfold3d(const float* x, int b, int c, int s, int e, float* rows, void* stream) {
    int rc = fold_check("rf_unfold3d", x, rows, b, c, s, e);
    if (rc) return rc;
    hipLaunchKernelGGL(k_unfold3d, dim3(grid_for((size_t)b * c * s * s * s)), dim3(256), 0, (hipStream_t)stream, x, b, c, s, e, rows);
    RF_CHECK_LAUNCH("rf_unfold3d");
    return RF_OK;
}

extern "C" int rf_fold3d(const float* rows, int b, int c, int s, int e, float* x, void* stream) {
    int rc = fold_check("rf_fold3d", rows, x, b, c, s, e);
    if (rc) return rc;
    hipLaunchKernelGGL(k_fold3d, dim3(grid_for((size_t)b * c * s * s * s)), dim3(256), 0, (hipStream_t)stream, rows, b, c, s, e, x);
    RF_CHECK_LAUNCH("rf_fold3d");
    return RF_OK;
}

// p_rows[(bb*r^3 + prow)][k][c][e^3]  <-  retrieved features of (bb, k) at the attention patch prow
__global__ __launch_bounds__(256) void k_attn_gather(const float* __restrict__ src, int layout, int b, int K, int c, int s, int e, int t,
                                                     float* __restrict__ p_rows) {
    const int r = s / e, q = s / t;
    const size_t e3 = (size_t)e * e * e, t3 = (size_t)t * t * t, r3 = (size_t)r * r * r;
    const size_t total = (size_t)b * r3 * K * c * e3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e2 = (int)(i % e), e1 = (int)((i / e) % e), e0 = (int)((i / ((size_t)e * e)) % e);
        const int cc = (int)((i / e3) % c);
        const int k = (int)((i / (e3 * c)) % K);
        const size_t row = i / (e3 * c * K);
        const int p2 = (int)(row % r), p1 = (int)((row / r) % r), p0 = (int)((row / ((size_t)r * r)) % r);
        const size_t bb = row / r3;
        const int d0 = p0 * e + e0, d1 = p1 * e + e1, d2 = p2 * e + e2;
        const size_t vol = bb * K + k;
        float v;
        if (layout == 0) {
            v = src[(((vol * c + cc) * s + d0) * s + d1) * s + d2];
        } else {
            const size_t patch = ((vol * q + d0 / t) * q + d1 / t) * q + d2 / t;
            v = src[(patch * c + cc) * t3 + ((size_t)(d0 % t) * t + (d1 % t)) * t + (d2 % t)];
        }
        p_rows[i] = v;
    }
}

extern "C" int rf_attn_gather_retrieved(const float* src, int src_layout, int b, int k, int c, int s, int e, int t,
                                        float* p_rows, void* stream) {
    RF_REQUIRE(src && p_rows && b > 0 && k > 0 && c > 0 && s > 0 && e > 0 && s % e == 0, RF_E_INVALID, "rf_attn_gather_retrieved: bad arguments");
    RF_REQUIRE(src_layout == 0 || (src_layout == 1 && t > 0 && s % t == 0 && t % e == 0), RF_E_INVALID,
               "rf_attn_gather_retrieved: bad layout %d / patch edge %d", src_layout, t);
    if (src_layout == 0) t = s;
    hipLaunchKernelGGL(k_attn_gather, dim3(grid_for((size_t)b * k * c * s * s * s)), dim3(256), 0, (hipStream_t)stream, src, src_layout, b, k, c,
                       s, e, t, p_rows);
    RF_CHECK_LAUNCH("rf_attn_gather_retrieved");
    return RF_OK;
}

// ------------------------------------------------------------------------------------------------- fused attention

__global__ __launch_bounds__(256) void k_attn_fuse(const float* __restrict__ x, const float* __restrict__ p, const float* __restrict__ xf,
                                                   const float* __restrict__ pf, const float* __restrict__ noise, int rows, int K, int d, int f,
                                                   int mode, float sharpness, float* __restrict__ out, float* __restrict__ scores_out,
                                                   float* __restrict__ weights_out) {
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        // ---- scores, switch, weights: every lane evaluates the row redundantly (uniform loads) with the arithmetic shared
        // with the volume-domain route (rf_attn_row_weights), so the two routes agree bit for bit
        float sc[RF_MAX_K], w[RF_MAX_K], sw;
        rf_attn_row_weights(xf + (size_t)row * f, pf + (size_t)row * K * f, noise ? noise + (size_t)row * K : nullptr, K, f, mode, sharpness, sc, w, sw);
        if (lane < K) {
            // static indexing only (runtime-indexed register arrays go to scratch)
            float sv = 0.f, wv = 0.f;
#pragma unroll
            for (int k = 0; k < RF_MAX_K; ++k)
                if (k == lane) { sv = sc[k]; wv = w[k]; }
            if (scores_out) scores_out[(size_t)row * K + lane] = sv;
            if (weights_out) weights_out[(size_t)row * K + lane] = wv;
        }

        // ---- weighted sum of the raw retrieved patches (g = Identity) and blend
        for (int j = lane; j < d; j += 64) {
            float ws = 0.f;
#pragma unroll
            for (int k = 0; k < RF_MAX_K; ++k)
                if (k < K) ws = fmaf(w[k], p[((size_t)row * K + k) * d + j], ws);
            const float xr = x[(size_t)row * d + j];
            out[(size_t)row * d + j] = __fadd_rn(__fmul_rn(xr, 1.f - sw), __fmul_rn(ws, sw));     // torch: x*(1-s) + p*s, no contraction
        }
    }
}

extern "C" int rf_attn_fuse(const float* x, const float* p, const float* xf, const float* pf, const float* noise,
                            int b, int k, int d, int f, int mode, float sharpness,
                            float* out, float* scores_out, float* weights_out, void* stream) {
    RF_REQUIRE(x && p && xf && pf && out && b > 0 && d > 0, RF_E_INVALID, "rf_attn_fuse: bad arguments");
    RF_REQUIRE(k >= 1 && k <= RF_MAX_K, RF_E_UNSUPPORTED, "rf_attn_fuse: K=%d outside 1..%d", k, RF_MAX_K);
    RF_REQUIRE(f >= 1 && f <= 128, RF_E_UNSUPPORTED, "rf_attn_fuse: feature width %d outside 1..128", f);
    RF_REQUIRE(mode == RF_ATTN_SOFTMAX || (mode == RF_ATTN_GUMBEL_HARD && noise), RF_E_INVALID,
               "rf_attn_fuse: Gumbel-hard mode needs the noise tensor");
    const int want = (b + 3) / 4;
    hipLaunchKernelGGL(k_attn_fuse, dim3(want < 4096 ? want : 4096), dim3(256), 0, (hipStream_t)stream, x, p, xf, pf, noise, b, k, d, f, mode,
                       sharpness, out, scores_out, weights_out);
    RF_CHECK_LAUNCH("rf_attn_fuse");
    return RF_OK;
}
