// Backward of one SingleConv 'gcr' layer  y = ReLU(conv3(GroupNorm(x)))  (reference model/unet.py:19-76), the minimal training
// slice of SURVEY.md section 8f row N4 (the reference trains through these layers, trainer/train_refinement.py:41-43,295-306):
//
//   dz        = dy * (y > 0)                                           rf_relu_backward
//   d xn      = conv3(dz, W^T with flipped taps), no ReLU              rf_conv3d_k3_gn (relu = 0) -- the forward MFMA kernel
//   dW        = sum_{n,v} dz[n,co,v] * xn[n,ci,v + tap - 1]            rf_conv3d_k3_wgrad         -- fp32 MFMA, this file
//   dx, dgamma, dbeta from d xn                                        rf_gn_backward             -- this file
//
// GroupNorm backward (biased variance, statistics over (C/G) x D x H x W per sample):  with xh = (x - mean) * rstd,
// g = d xn * gamma_c, m = cpg * vol:   dx = rstd * (g - mean_group(g) - xh * mean_group(g * xh)),
// dgamma_c = sum_{n,v} d xn * xh,  dbeta_c = sum_{n,v} d xn.  Sums in float64, fixed order (deterministic).
#include "common.h"

// ---------------------------------------------------------------------------------------------------- ReLU mask
__global__ __launch_bounds__(256) void k_relu_bwd(const float4* __restrict__ dy, const float4* __restrict__ y, size_t n4, float4* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 g = dy[i], v = y[i];
        out[i] = make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f);
    }
}

extern "C" int rf_relu_backward(const float* dy, const float* y, size_t count, float* out, void* stream) {
    RF_REQUIRE(dy && y && out && count > 0 && (count & 3) == 0, RF_E_INVALID, "rf_relu_backward: bad arguments (count must be a multiple of 4)");
    const size_t want = (count / 4 + 255) / 256;
    hipLaunchKernelGGL(k_relu_bwd, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, (hipStream_t)stream, (const float4*)dy, (const float4*)y,
                       count / 4, (float4*)out);
    RF_CHECK_LAUNCH("rf_relu_backward");
    return RF_OK;
}

// ------------------------------------------------------------------------------------------- GroupNorm backward
// pass 1: per (sample, channel): mean / rstd of its group (recomputed: sum, sum of squares over the group), then
// a1 = sum_v dxn, a2 = sum_v dxn * xh.  One workgroup per (n, c); every workgroup of a group recomputes the group moments
// (cpg <= 16 re-reads of x: the backward pass is not the hot path).
__global__ __launch_bounds__(256) void k_gnb_reduce(const float* __restrict__ x, const float* __restrict__ dxn, int C, int cpg, size_t vol, double eps,
                                                    double2* __restrict__ moments, double* __restrict__ a1_out, double* __restrict__ a2_out) {
    const int nc = blockIdx.x, nn = nc / C, c = nc % C, g = c / cpg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ double red[8];
    const float* xg = x + ((size_t)nn * C + (size_t)g * cpg) * vol;
    const size_t glen = (size_t)cpg * vol;
    double s = 0.0, q = 0.0;
    if ((vol & 3) == 0) {                                           // 16-byte loads (every volume but 1^3)
        const float4* xg4 = reinterpret_cast<const float4*>(xg);
        for (size_t i = tid; i < glen / 4; i += 256) {
            const float4 t = xg4[i];
            const double v0 = t.x, v1 = t.y, v2 = t.z, v3 = t.w;
            s += (v0 + v1) + (v2 + v3); q += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
        }
    } else {
        for (size_t i = tid; i < glen; i += 256) { const double v = xg[i]; s += v; q += v * v; }
    }
    s = wave_sum(s); q = wave_sum(q);
    if (lane == 0) { red[wave * 2] = s; red[wave * 2 + 1] = q; }
    __syncthreads();
    double S = 0.0, Q = 0.0;
    for (int w = 0; w < 4; ++w) { S += red[w * 2]; Q += red[w * 2 + 1]; }
    __syncthreads();
    const double mean = S / (double)glen;
    double var = Q / (double)glen - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + eps);
    const float* xc = x + (size_t)nc * vol;
    const float* dc = dxn + (size_t)nc * vol;
    double a1 = 0.0, a2 = 0.0;
    if ((vol & 3) == 0) {
        const float4* xc4 = reinterpret_cast<const float4*>(xc);
        const float4* dc4 = reinterpret_cast<const float4*>(dc);
        for (size_t i = tid; i < vol / 4; i += 256) {
            const float4 xv = xc4[i], dv = dc4[i];
            const double d0 = dv.x, d1 = dv.y, d2 = dv.z, d3 = dv.w;
            a1 += (d0 + d1) + (d2 + d3);
            a2 += (d0 * (((double)xv.x - mean) * rstd) + d1 * (((double)xv.y - mean) * rstd)) +
                  (d2 * (((double)xv.z - mean) * rstd) + d3 * (((double)xv.w - mean) * rstd));
        }
    } else {
        for (size_t i = tid; i < vol; i += 256) { const double d = dc[i]; a1 += d; a2 += d * ((double)xc[i] - mean) * rstd; }
    }
    a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (lane == 0) { red[wave * 2] = a1; red[wave * 2 + 1] = a2; }
    __syncthreads();
    if (tid == 0) {
        double A1 = 0.0, A2 = 0.0;
        for (int w = 0; w < 4; ++w) { A1 += red[w * 2]; A2 += red[w * 2 + 1]; }
        a1_out[nc] = A1;
        a2_out[nc] = A2;
        moments[nc] = make_double2(mean, rstd);
    }
}

// pass 2: dx; and per (n, c) the pieces of dgamma / dbeta (summed over n by the caller in float64)
__global__ __launch_bounds__(256) void k_gnb_apply(const float* __restrict__ x, const float* __restrict__ dxn, const float* __restrict__ gamma, int C,
                                                   int cpg, size_t vol, const double2* __restrict__ moments, const double* __restrict__ a1,
                                                   const double* __restrict__ a2, float* __restrict__ dx) {
    const int nc = blockIdx.x, nn = nc / C, c = nc % C, g = c / cpg;
    double G1 = 0.0, G2 = 0.0;                                       // sum over the group of gamma * a1, gamma * a2
    for (int k = 0; k < cpg; ++k) {
        const int cc = g * cpg + k;
        G1 += (double)gamma[cc] * a1[(size_t)nn * C + cc];
        G2 += (double)gamma[cc] * a2[(size_t)nn * C + cc];
    }
    const double2 mo = moments[nc];
    const double m = (double)cpg * (double)vol;
    const double k0 = mo.y * (double)gamma[c], k1 = mo.y * G1 / m, k2 = mo.y * G2 / m;
    const float* xc = x + (size_t)nc * vol;
    const float* dc = dxn + (size_t)nc * vol;
    float* o = dx + (size_t)nc * vol;
    if ((vol & 3) == 0) {
        const float4* xc4 = reinterpret_cast<const float4*>(xc);
        const float4* dc4 = reinterpret_cast<const float4*>(dc);
        float4* o4 = reinterpret_cast<float4*>(o);
        for (size_t i = (size_t)blockIdx.y * 256 + threadIdx.x; i < vol / 4; i += (size_t)gridDim.y * 256) {
            const float4 xv = xc4[i], dv = dc4[i];
            float4 r;
            r.x = (float)(k0 * (double)dv.x - k1 - (((double)xv.x - mo.x) * mo.y) * k2);
            r.y = (float)(k0 * (double)dv.y - k1 - (((double)xv.y - mo.x) * mo.y) * k2);
            r.z = (float)(k0 * (double)dv.z - k1 - (((double)xv.z - mo.x) * mo.y) * k2);
            r.w = (float)(k0 * (double)dv.w - k1 - (((double)xv.w - mo.x) * mo.y) * k2);
            o4[i] = r;
        }
    } else {
        for (size_t i = (size_t)blockIdx.y * 256 + threadIdx.x; i < vol; i += (size_t)gridDim.y * 256) {
            const double xh = ((double)xc[i] - mo.x) * mo.y;
            o[i] = (float)(k0 * (double)dc[i] - k1 - xh * k2);
        }
    }
}

extern "C" size_t rf_gn_backward_ws_bytes(int n, int c) { return (size_t)n * c * sizeof(double2); }

// x, dxn [n][c][edge^3]; gamma [c]; out: dx [n][c][edge^3], dgamma_parts / dbeta_parts [n][c] float64 (the caller sums over n)
extern "C" int rf_gn_backward(const float* x, const float* dxn, int n, int c, int edge, const float* gamma, int groups, float eps, float* dx,
                              double* dgamma_parts, double* dbeta_parts, void* ws, size_t ws_bytes, void* stream) {
    RF_REQUIRE(x && dxn && gamma && dx && dgamma_parts && dbeta_parts && ws && n > 0 && c > 0 && edge > 0 && groups > 0 && c % groups == 0, RF_E_INVALID,
               "rf_gn_backward: bad arguments");
    RF_REQUIRE(ws_bytes >= rf_gn_backward_ws_bytes(n, c), RF_E_WORKSPACE, "rf_gn_backward: workspace too small");
    const size_t vol = (size_t)edge * edge * edge;
    const int cpg = c / groups;
    double2* moments = (double2*)ws;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_gnb_reduce, dim3(n * c), dim3(256), 0, s, x, dxn, c, cpg, vol, (double)eps, moments, dbeta_parts, dgamma_parts);
    RF_CHECK_LAUNCH("rf_gn_backward(reduce)");
    const unsigned gy = (unsigned)((vol / 4 + 1023) / 1024);          // 1024 float4 per workgroup
    hipLaunchKernelGGL(k_gnb_apply, dim3(n * c, gy < 1 ? 1 : gy), dim3(256), 0, s, x, dxn, gamma, c, cpg, vol, moments, dbeta_parts, dgamma_parts, dx);
    RF_CHECK_LAUNCH("rf_gn_backward(apply)");
    return RF_OK;
}

// ------------------------------------------------------------------------------------------------- weight gradient
// dW[co][ci][tap] = sum over samples and voxels of dz[n][co][v] * xn[n][ci][v + tap - 1]   (xn = GroupNorm(x), zero padded)
// as an fp32-MFMA GEMM per workgroup:  D[co][col] += sum_k A[co][k] * B[k][col]  with k = voxels of an 8^3 box (4 per MFMA step),
// M = a 16-cout block, N = the 4 channels x 27 taps of one channel chunk (108 columns -> 7 n-blocks of 16, one per wave; wave 7
// helps staging only).  A = dz tile [16][512] in LDS, B = the GroupNorm-applied halo box [4][10^3] in LDS read through a per-lane
// column offset (channel * 1000 + tap offset).  A workgroup walks the boxes b = g, g + GB, ... of its (channel chunk, cout block)
// and keeps its 16 x 108 partial in registers; partials [GB][cout][cin][27] are then reduced in float64 in a fixed order.
// Edge >= 8 volumes only; 4^3 / 2^3 / 1^3 layers go through rf_linear on the unfolded input (rfuse/autograd.py).
struct WgradArgs {
    const float* x;
    const float4* affine;
    const float* dz;
    float* parts;             // [GB][cout][cin][27]
    int cin, cout, n, edge, gb;
};

__global__ __launch_bounds__(512) void k_conv3_wgrad(WgradArgs a) {
    constexpr int HE = 10, CH = HE * HE * HE, DZS = 512 + 4;
    __shared__ float xs[4 * CH];
    __shared__ __attribute__((aligned(16))) float ds[16 * DZS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int cchunk = blockIdx.x, cob = blockIdx.y * 16, g = blockIdx.z;
    const int edge = a.edge, bpe = edge / 8, boxes_per_sample = bpe * bpe * bpe;
    const int nboxes = a.n * boxes_per_sample;
    const size_t vol = (size_t)edge * edge * edge;
    // this lane's B column: n-block = wave, column c = wave*16 + li -> (channel, tap)
    const int col = wave * 16 + li;
    const bool col_ok = wave < 7 && col < 108;
    const int cch = col_ok ? col / 27 : 0, ctap = col_ok ? col % 27 : 0;
    const int coloff = cch * CH + ((ctap / 9) * HE + (ctap / 3) % 3) * HE + ctap % 3;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int b = g; b < nboxes; b += a.gb) {
        const int nn = b / boxes_per_sample, bb = b % boxes_per_sample;
        const int x0 = (bb % bpe) * 8, y0 = ((bb / bpe) % bpe) * 8, z0 = (bb / (bpe * bpe)) * 8;
        __syncthreads();                                            // previous box fully consumed
        for (int i = tid; i < 4 * CH; i += 512) {                   // GroupNorm-applied halo box, zero outside the volume / past cin
            const int c = i / CH, r = i % CH;
            const int hx = r % HE, hy = (r / HE) % HE, hz = r / (HE * HE);
            const int z = z0 + hz - 1, y = y0 + hy - 1, x = x0 + hx - 1, ci = cchunk * 4 + c;
            float v = 0.f;
            if (ci < a.cin && (unsigned)z < (unsigned)edge && (unsigned)y < (unsigned)edge && (unsigned)x < (unsigned)edge) {
                const float4 af = a.affine[(size_t)nn * a.cin + ci];
                v = fmaf(a.x[((size_t)nn * a.cin + ci) * vol + ((size_t)z * edge + y) * edge + x] - af.x, af.y, af.z);
            }
            xs[i] = v;
        }
        for (int i = tid; i < 16 * 512; i += 512) {                 // dz tile [16 couts][512 voxels of the box]
            const int co = i >> 9, v = i & 511;
            const int x = v & 7, y = (v >> 3) & 7, z = v >> 6;
            float d = 0.f;
            if (cob + co < a.cout) d = a.dz[((size_t)nn * a.cout + cob + co) * vol + ((size_t)(z0 + z) * edge + (y0 + y)) * edge + (x0 + x)];
            ds[co * DZS + v] = d;
        }
        __syncthreads();
        if (wave < 7) {
#pragma unroll 4
            for (int v0 = 0; v0 < 512; v0 += 4) {
                const int v = v0 + kq;                              // this lane's voxel of the k-step
                const int hb = (((v >> 6)) * HE + ((v >> 3) & 7)) * HE + (v & 7);
                const float av = ds[li * DZS + v];
                const float bv = xs[coloff + hb];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
            }
        }
    }
    // D[row = co (4*kq + r)][col = li]
    if (col_ok) {
        const int ci = cchunk * 4 + cch;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = cob + kq * 4 + r;
            if (co < a.cout && ci < a.cin) a.parts[(((size_t)g * a.cout + co) * a.cin + ci) * 27 + ctap] = acc[r];
        }
    }
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ parts, int gb, size_t count, float* __restrict__ dw) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int g = 0; g < gb; ++g) s += (double)parts[(size_t)g * count + i];
        dw[i] = (float)s;
    }
}

static int wgrad_groups(int n, int edge) {
    const long long boxes = (long long)n * (edge / 8) * (edge / 8) * (edge / 8);
    return (int)(boxes < 256 ? boxes : 256);         // workgroups per (channel chunk, cout block): 64 left most CUs with one 8-wave workgroup
}

extern "C" size_t rf_conv3d_k3_wgrad_ws_bytes(int cin, int cout, int n, int edge) {
    return (size_t)wgrad_groups(n, edge) * cout * cin * 27 * sizeof(float);
}

// x [n][cin][edge^3] (the layer input, single source), gn_affine as the forward, dz [n][cout][edge^3] -> dw OIDHW [cout][cin][27]
extern "C" int rf_conv3d_k3_wgrad(const float* x, int cin, int n, int edge, const float* gn_affine, const float* dz, int cout, float* dw, void* ws,
                                  size_t ws_bytes, void* stream) {
    RF_REQUIRE(x && gn_affine && dz && dw && ws && cin > 0 && cout > 0 && n > 0, RF_E_INVALID, "rf_conv3d_k3_wgrad: bad arguments");
    RF_REQUIRE(rf_is_pow2(edge) && edge >= 8 && edge <= 128, RF_E_UNSUPPORTED, "rf_conv3d_k3_wgrad: edge %d (8^3 boxes: a power of two >= 8)", edge);
    RF_REQUIRE(ws_bytes >= rf_conv3d_k3_wgrad_ws_bytes(cin, cout, n, edge), RF_E_WORKSPACE, "rf_conv3d_k3_wgrad: workspace too small");
    WgradArgs a;
    a.x = x; a.affine = reinterpret_cast<const float4*>(gn_affine); a.dz = dz; a.parts = (float*)ws;
    a.cin = cin; a.cout = cout; a.n = n; a.edge = edge; a.gb = wgrad_groups(n, edge);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_conv3_wgrad, dim3((cin + 3) / 4, (cout + 15) / 16, a.gb), dim3(512), 0, s, a);
    RF_CHECK_LAUNCH("rf_conv3d_k3_wgrad");
    const size_t count = (size_t)cout * cin * 27;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, (const float*)ws, a.gb, count, dw);
    RF_CHECK_LAUNCH("rf_conv3d_k3_wgrad(reduce)");
    return RF_OK;
}
