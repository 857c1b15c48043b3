// Dense layers for gfx950: y = act(x . W^T + b) as an fp32-MFMA GEMM (v_mfma_f32_16x16x4_f32), plus row L2 normalise.
//
// Reference arithmetic being replaced: nn.Linear + LeakyReLU(0.01) stacks of AttentionFeatureEncoder
// (model/attention.py:36-42), nn.Linear + ReLU stacks of Patch04 (model/retrieval.py:68-78), final_layer of the conv
// patch encoders (model/retrieval.py:149), F.normalize (util/retrieval.py:66).
//
// Tile: one workgroup (4 waves) = 128 rows x NB*16 columns; wave w owns rows [32w, 32w+32) (MB = 2 accumulator row
// blocks) and all NB column blocks.  K is walked in chunks of 32: the x tile [128][32] and the packed weight slab
// [32][NB*16] are staged in LDS.  fp32-input MFMA == fp32 FMA chain in k order, so results are plain fp32.
#include "common.h"

__global__ void k_linear_pack(const float* __restrict__ w, int nout, int nin, int nin4, int nout16, float* __restrict__ wp) {
    const size_t total = (size_t)nin4 * nout16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % nout16), k = (int)(i / nout16);
        wp[i] = (co < nout && k < nin) ? w[(size_t)co * nin + k] : 0.f;
    }
}

extern "C" size_t rf_linear_packed_floats(int nout, int nin) { return (size_t)rf_round_up(nin, 4) * rf_round_up(nout, 16); }

extern "C" int rf_linear_pack_weight(const float* w, int nout, int nin, float* wp, void* stream) {
    RF_REQUIRE(w && wp && nout > 0 && nin > 0, RF_E_INVALID, "rf_linear_pack_weight: bad arguments");
    const size_t total = rf_linear_packed_floats(nout, nin);
    const size_t want = (total + 255) / 256;
    hipLaunchKernelGGL(k_linear_pack, dim3((unsigned)(want < 1024 ? want : 1024)), dim3(256), 0, (hipStream_t)stream, w, nout, nin,
                       rf_round_up(nin, 4), rf_round_up(nout, 16), wp);
    RF_CHECK_LAUNCH("rf_linear_pack_weight");
    return RF_OK;
}

template <int NB>
__global__ __launch_bounds__(256, 2) void k_linear_mfma(const float* __restrict__ x, int rows, int nin, const float* __restrict__ wp,
                                                        int nin4, int nout16, const float* __restrict__ bias, int nout, int act, float slope,
                                                        float* __restrict__ y) {
    constexpr int BM = 128, KC = 32, XSTR = KC + 4, NCO = NB * 16, COS = NCO + ((NB % 2 == 0) ? 16 : 0), MB = 2;
    __shared__ __attribute__((aligned(16))) float xs[BM * XSTR];
    __shared__ __attribute__((aligned(16))) float ws[KC * COS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * BM, cob = blockIdx.y * NCO;
    const bool vec_ok = (nin & 3) == 0;

    f32x4 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int aoff[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) aoff[mb] = (wave * 32 + mb * 16 + (lane & 15)) * XSTR + (lane >> 4);
    const int boff = (lane >> 4) * COS + (lane & 15);

    for (int k0 = 0; k0 < nin4; k0 += KC) {
        // x tile: 128 rows x 8 float4
        for (int i = tid; i < BM * (KC / 4); i += 256) {
            const int r = i / (KC / 4), k4 = i % (KC / 4);
            const int row = row0 + r, k = k0 + k4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < rows) {
                const float* p = x + (size_t)row * nin + k;
                if (vec_ok) {
                    if (k < nin) v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (k + 0 < nin) v.x = p[0];
                    if (k + 1 < nin) v.y = p[1];
                    if (k + 2 < nin) v.z = p[2];
                    if (k + 3 < nin) v.w = p[3];
                }
            }
            *reinterpret_cast<float4*>(xs + r * XSTR + k4 * 4) = v;
        }
        // weight slab: 32 k-rows x NCO/4 float4
        for (int i = tid; i < KC * (NCO / 4); i += 256) {
            const int kr = i / (NCO / 4), co4 = i % (NCO / 4);
            const int k = k0 + kr, co = cob + co4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < nin4 && co < nout16) v = *reinterpret_cast<const float4*>(wp + (size_t)k * nout16 + co);
            *reinterpret_cast<float4*>(ws + kr * COS + co4 * 4) = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KC / 4; ++kk) {
            float av[MB], bv[NB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) av[mb] = xs[aoff[mb] + kk * 4];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bv[nb] = ws[boff + kk * 4 * COS + nb * 16];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], bv[nb], acc[mb][nb], 0, 0, 0);
        }
        __syncthreads();
    }

#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int co = cob + nb * 16 + (lane & 15);
        if (co >= nout) continue;
        const float bv = bias ? bias[co] : 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + wave * 32 + mb * 16 + (lane >> 4) * 4 + r;
                if (row < rows) {
                    float v = acc[mb][nb][r] + bv;
                    if (act == RF_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (act == RF_ACT_LEAKY) v = v > 0.f ? v : v * slope;
                    y[(size_t)row * nout + co] = v;
                }
            }
        }
    }
}

extern "C" int rf_linear(const float* x, int rows, int nin, const float* w_packed, const float* bias, int nout,
                         int act, float slope, float* y, void* stream) {
    RF_REQUIRE(x && w_packed && y && rows > 0 && nin > 0 && nout > 0, RF_E_INVALID, "rf_linear: bad arguments");
    RF_REQUIRE(act >= RF_ACT_NONE && act <= RF_ACT_LEAKY, RF_E_INVALID, "rf_linear: unknown activation %d", act);
    const int nin4 = rf_round_up(nin, 4), nout16 = rf_round_up(nout, 16);
    hipStream_t s = (hipStream_t)stream;
    const unsigned gx = (unsigned)((rows + 127) / 128);
    if (nout16 <= 32) {
        hipLaunchKernelGGL(k_linear_mfma<2>, dim3(gx, (nout16 + 31) / 32), dim3(256), 0, s, x, rows, nin, w_packed, nin4, nout16, bias, nout, act,
                           slope, y);
    } else {
        hipLaunchKernelGGL(k_linear_mfma<4>, dim3(gx, (nout16 + 63) / 64), dim3(256), 0, s, x, rows, nin, w_packed, nin4, nout16, bias, nout, act,
                           slope, y);
    }
    RF_CHECK_LAUNCH("rf_linear");
    return RF_OK;
}

// ---------------------------------------------------------------------------------------------------- weight gradient
// dW[m][n] = sum_k a[k][m] * b[k][n]: the weight gradient of a Linear layer (a = dL/d(pre-activation) [K rows][M = nout],
// b = the layer input [K rows][N = nin]; rfuse/autograd.py, reference trainer/train_refinement.py:108-116 trains through these).
// K is the row count of the batch (10^4..10^6), M and N are layer widths (<= 512): a split-K GEMM.  Both operands are already in
// MFMA operand order -- lane l of an A read takes a[k0 + l/16][m0 + l%16], of a B read b[k0 + l/16][n0 + l%16], i.e. 16 consecutive
// floats of 4 consecutive rows -- so they go global -> VGPR -> MFMA directly, no LDS.  A workgroup owns a 32 x 64 tile of dW and one
// K slice; its 4 waves interleave the slice's k-steps, reduce through LDS, and write an fp32 partial; k_linear_wgrad_reduce sums the
// slices in float64 in a fixed order (deterministic, no atomics).
__global__ __launch_bounds__(256) void k_linear_wgrad(const float* __restrict__ a, const float* __restrict__ b, int K, int M, int N, int kslice,
                                                     float* __restrict__ partial) {
    constexpr int MB = 2, NB = 4, U = 4;                            // k-steps in flight per wave
    __shared__ float red[3][32 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 64;
    const int kbeg = blockIdx.z * kslice, kend = min(K, kbeg + kslice);
    const int kq = lane >> 4, li = lane & 15;
    f32x4 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bool mok[MB], nok[NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) mok[mb] = m0 + mb * 16 + li < M;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) nok[nb] = n0 + nb * 16 + li < N;
    for (int k0 = kbeg + wave * 4 * U; k0 < kend; k0 += 16 * U) {
        float av[U][MB], bv[U][NB];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * 4 + kq;
            const bool kok = k < kend;
            const float* ar = a + (size_t)k * M + m0 + li;
            const float* br = b + (size_t)k * N + n0 + li;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) av[u][mb] = (kok && mok[mb]) ? ar[mb * 16] : 0.f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bv[u][nb] = (kok && nok[nb]) ? br[nb * 16] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][mb], bv[u][nb], acc[mb][nb], 0, 0, 0);
    }
    // D rows 4*kq + r (m), column li (n): waves 1..3 -> LDS, wave 0 adds them in wave order
    if (wave > 0) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave - 1][(mb * 16 + kq * 4 + r) * 64 + nb * 16 + li] = acc[mb][nb][r];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = partial + (size_t)blockIdx.z * M * N;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ml = mb * 16 + kq * 4 + r, nl = nb * 16 + li;
                    const float v = ((acc[mb][nb][r] + red[0][ml * 64 + nl]) + red[1][ml * 64 + nl]) + red[2][ml * 64 + nl];
                    if (m0 + ml < M && n0 + nl < N) out[(size_t)(m0 + ml) * N + n0 + nl] = v;
                }
    }
}

__global__ void k_linear_wgrad_reduce(const float* __restrict__ partial, int slices, size_t mn, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < mn; i += (size_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int z = 0; z < slices; ++z) s += (double)partial[(size_t)z * mn + i];
        out[i] = (float)s;
    }
}

static void linear_wgrad_plan(int K, int M, int N, int& slices, int& kslice) {
    const int tiles = ((M + 31) / 32) * ((N + 63) / 64);
    int want = (1024 + tiles - 1) / tiles;                           // ~4 workgroups per CU
    const int maxs = (K + 255) / 256;                               // at least 256 rows (16 k-steps per wave) per slice
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    kslice = ((K + want - 1) / want + 15) / 16 * 16;
    slices = (K + kslice - 1) / kslice;
}

extern "C" size_t rf_linear_wgrad_ws_bytes(int K, int M, int N) {
    int slices, kslice;
    linear_wgrad_plan(K, M, N, slices, kslice);
    return (size_t)slices * M * N * sizeof(float);
}

extern "C" int rf_linear_wgrad(const float* a, const float* b, int K, int M, int N, float* dw, void* ws, size_t ws_bytes, void* stream) {
    RF_REQUIRE(a && b && dw && ws && K > 0 && M > 0 && N > 0, RF_E_INVALID, "rf_linear_wgrad: bad arguments");
    RF_REQUIRE(ws_bytes >= rf_linear_wgrad_ws_bytes(K, M, N), RF_E_INVALID, "rf_linear_wgrad: workspace of %zu bytes is too small", ws_bytes);
    int slices, kslice;
    linear_wgrad_plan(K, M, N, slices, kslice);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_linear_wgrad, dim3((M + 31) / 32, (N + 63) / 64, slices), dim3(256), 0, s, a, b, K, M, N, kslice, static_cast<float*>(ws));
    RF_CHECK_LAUNCH("rf_linear_wgrad");
    const size_t mn = (size_t)M * N;
    hipLaunchKernelGGL(k_linear_wgrad_reduce, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, s, static_cast<const float*>(ws), slices, mn, dw);
    RF_CHECK_LAUNCH("rf_linear_wgrad(reduce)");
    return RF_OK;
}

// one wave per row
__global__ __launch_bounds__(256) void k_l2norm_rows(float* __restrict__ x, int rows, int dim, float eps) {
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        float* p = x + (size_t)row * dim;
        float s = 0.f;
        for (int j = lane; j < dim; j += 64) s += p[j] * p[j];
        s = wave_sum(s);
        const float denom = fmaxf(sqrtf(s), eps);
        for (int j = lane; j < dim; j += 64) p[j] = p[j] / denom;
    }
}

extern "C" int rf_l2_normalize_rows(float* x, int rows, int dim, float eps, void* stream) {
    RF_REQUIRE(x && rows > 0 && dim > 0, RF_E_INVALID, "rf_l2_normalize_rows: bad arguments");
    const int want = (rows + 3) / 4;
    hipLaunchKernelGGL(k_l2norm_rows, dim3(want < 2048 ? want : 2048), dim3(256), 0, (hipStream_t)stream, x, rows, dim, eps);
    RF_CHECK_LAUNCH("rf_l2_normalize_rows");
    return RF_OK;
}
