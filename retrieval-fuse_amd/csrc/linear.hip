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
