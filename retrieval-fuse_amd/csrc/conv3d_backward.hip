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

// The same mask, and max |out| beside it: workgroup b leaves its maximum in slot b of RF_AMAX_SLOTS floats (and zeroes the slots b + gridDim.x, ...
// nobody owns), the consumer takes the maximum of the slots.  (One atomicMax per workgroup on a single word was 10 ns each: 0.16 ms for 16 384
// workgroups beside 0.03 ms of copying.)  The data-gradient conv on the F16 matrix cores wants its input scaled into the split forms' range.
namespace { constexpr int RF_AMAX_SLOTS = 1024; }

__global__ __launch_bounds__(256) void k_relu_bwd_amax(const float4* __restrict__ dy, const float4* __restrict__ y, size_t n4, float4* __restrict__ out,
                                                       float* __restrict__ slots) {
    float m = 0.f;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 g = dy[i], v = y[i];
        const float4 o = make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f);
        out[i] = o;
        m = fmaxf(fmaxf(m, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        // fmaxf drops a NaN and the split forms clamp: an upstream gradient that is not finite must stay visible (overflow / anomaly detection of the
        // training loop), so it is recorded as an infinite maximum, which rf_dgrad_scale_affine turns into NaN gradients
        bad = bad || !(fabsf(o.x) < INFINITY) || !(fabsf(o.y) < INFINITY) || !(fabsf(o.z) < INFINITY) || !(fabsf(o.w) < INFINITY);
    }
    if (bad) m = INFINITY;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        slots[blockIdx.x] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        for (unsigned k = blockIdx.x + gridDim.x; k < (unsigned)RF_AMAX_SLOTS; k += gridDim.x) slots[k] = 0.f;
    }
}

extern "C" int rf_relu_backward_amax_slots(void) { return RF_AMAX_SLOTS; }

extern "C" int rf_relu_backward_amax(const float* dy, const float* y, size_t count, float* out, float* amax_slots, void* stream) {
    RF_REQUIRE(dy && y && out && amax_slots && count > 0 && (count & 3) == 0, RF_E_INVALID, "rf_relu_backward_amax: bad arguments (count must be a multiple of 4)");
    const size_t want = (count / 4 + 255) / 256;
    hipLaunchKernelGGL(k_relu_bwd_amax, dim3((unsigned)(want < (size_t)RF_AMAX_SLOTS ? want : (size_t)RF_AMAX_SLOTS)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)dy, (const float4*)y, count / 4, (float4*)out, amax_slots);
    RF_CHECK_LAUNCH("rf_relu_backward_amax");
    return RF_OK;
}

// identity GroupNorm affine (mean 0, shift 0) with scale s = the power of two that puts max |dz| into [512, 1024): rows x (0, s, 0, 0), and
// scales = (s, 1 / s).  A zero maximum gives s = 1; a maximum that is not finite (rf_relu_backward_amax records inf / NaN gradients as +inf) gives s = 1 and
// 1 / s = NaN: the data gradient (rf_gn_backward's dxn_inv_scale) and the weight gradient (the split wgrad's reduction) are multiplied by it and come out
// NaN everywhere -- the fp32 route and the reference propagate inf / NaN, and a clamped, finite gradient would hide an overflow from the training loop.
// Everything stays on the device: no host sync in the backward pass.
__global__ __launch_bounds__(256) void k_dgrad_affine(const float* __restrict__ slots, int rows, float4* __restrict__ affine, float* __restrict__ scales) {
    float m = 0.f;
    for (int k = threadIdx.x; k < RF_AMAX_SLOTS; k += 256) m = fmaxf(m, slots[k]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    float s = 1.f;
    if (m > 0.f && m < INFINITY) {
        int e = 9 - ilogbf(m);
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
        s = ldexpf(1.f, e);
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < rows) affine[i] = make_float4(0.f, s, 0.f, 0.f);
    if (i == 0) { scales[0] = s; scales[1] = (m < INFINITY) ? 1.f / s : __builtin_nanf(""); }
}

extern "C" int rf_dgrad_scale_affine(const float* amax_slots, int rows, float* affine, float* scales, void* stream) {
    RF_REQUIRE(amax_slots && affine && scales && rows > 0, RF_E_INVALID, "rf_dgrad_scale_affine: bad arguments");
    hipLaunchKernelGGL(k_dgrad_affine, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, amax_slots, rows, reinterpret_cast<float4*>(affine), scales);
    RF_CHECK_LAUNCH("rf_dgrad_scale_affine");
    return RF_OK;
}

// ------------------------------------------------------------------------------------- pooling / upsampling of the training slice
// MaxPool3d(2) backward (reference model/unet.py:159 pools between the encoder levels): the gradient of a pooled cell goes to the cell's FIRST
// maximum in (z, y, x) order -- torch.nn.MaxPool3d's choice (its scan keeps the first of equal values), the other seven voxels get 0.
// One thread per pooled cell: 4 float2 loads of x, 4 float2 stores.  rows = n * c volumes of edge^3.
__global__ __launch_bounds__(256) void k_maxpool2_bwd(const float* __restrict__ x, const float* __restrict__ dy, size_t cells, int half, float* __restrict__ dx) {
    const int edge = 2 * half;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < cells; i += (size_t)gridDim.x * 256) {
        const size_t row = i / ((size_t)half * half * half);
        const int q = (int)(i % ((size_t)half * half * half));
        const int px = q % half, py = (q / half) % half, pz = q / (half * half);
        const size_t base = row * (size_t)edge * edge * edge + ((size_t)(2 * pz) * edge + 2 * py) * edge + 2 * px;
        float2 v[4];
        int best = 0;
        float m = -INFINITY;                                        // torch's scan: if (val > max || isnan(val)) { max = val; index = here; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = *reinterpret_cast<const float2*>(x + base + ((size_t)(k >> 1) * edge + (k & 1)) * edge);
            if (v[k].x > m || v[k].x != v[k].x) { m = v[k].x; best = 2 * k; }
            if (v[k].y > m || v[k].y != v[k].y) { m = v[k].y; best = 2 * k + 1; }
        }
        const float g = dy[i];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            *reinterpret_cast<float2*>(dx + base + ((size_t)(k >> 1) * edge + (k & 1)) * edge) = make_float2(best == 2 * k ? g : 0.f, best == 2 * k + 1 ? g : 0.f);
    }
}

extern "C" int rf_maxpool3d_2_backward(const float* x, const float* dy, int n, int c, int edge, float* dx, void* stream) {
    RF_REQUIRE(x && dy && dx && n > 0 && c > 0 && edge >= 2 && edge % 2 == 0, RF_E_INVALID, "rf_maxpool3d_2_backward: bad arguments");
    const int half = edge / 2;
    const size_t cells = (size_t)n * c * half * half * half, want = (cells + 255) / 256;
    hipLaunchKernelGGL(k_maxpool2_bwd, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0, (hipStream_t)stream, x, dy, cells, half, dx);
    RF_CHECK_LAUNCH("rf_maxpool3d_2_backward");
    return RF_OK;
}

// nearest-neighbour x2 upsample (reference model/unet.py:297-308: F.interpolate(scale_factor = 2) ahead of a decoder's concat) and its backward, the sum
// over each 2x2x2 cell in (z, y, x) order.  One thread per low-resolution voxel pair / voxel; rows = n * c volumes.
__global__ __launch_bounds__(256) void k_upsample2(const float* __restrict__ lo, size_t pairs, int e, float* __restrict__ hi) {
    const int hp = e / 2, E = 2 * e;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < pairs; i += (size_t)gridDim.x * 256) {
        const size_t row = i / ((size_t)hp * e * e);
        const int q = (int)(i % ((size_t)hp * e * e));
        const int xp = q % hp, y = (q / hp) % e, z = q / (hp * e);
        const float2 v = *reinterpret_cast<const float2*>(lo + row * (size_t)e * e * e + ((size_t)z * e + y) * e + 2 * xp);
        const float4 o = make_float4(v.x, v.x, v.y, v.y);
        float* dst = hi + row * (size_t)E * E * E + ((size_t)(2 * z) * E + 2 * y) * E + 4 * xp;
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(dst + ((size_t)(k >> 1) * E + (k & 1)) * E) = o;
    }
}

extern "C" int rf_upsample3d_2(const float* lo, int n, int c, int edge_lo, float* hi, void* stream) {
    RF_REQUIRE(lo && hi && n > 0 && c > 0 && edge_lo >= 2 && edge_lo % 2 == 0, RF_E_INVALID, "rf_upsample3d_2: bad arguments (even low-resolution edge)");
    const size_t pairs = (size_t)n * c * edge_lo * edge_lo * (edge_lo / 2), want = (pairs + 255) / 256;
    hipLaunchKernelGGL(k_upsample2, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0, (hipStream_t)stream, lo, pairs, edge_lo, hi);
    RF_CHECK_LAUNCH("rf_upsample3d_2");
    return RF_OK;
}

__global__ __launch_bounds__(256) void k_sumpool2(const float* __restrict__ hi, size_t cells, int half, float* __restrict__ lo) {
    const int edge = 2 * half;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < cells; i += (size_t)gridDim.x * 256) {
        const size_t row = i / ((size_t)half * half * half);
        const int q = (int)(i % ((size_t)half * half * half));
        const int px = q % half, py = (q / half) % half, pz = q / (half * half);
        const size_t base = row * (size_t)edge * edge * edge + ((size_t)(2 * pz) * edge + 2 * py) * edge + 2 * px;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float2 v = *reinterpret_cast<const float2*>(hi + base + ((size_t)(k >> 1) * edge + (k & 1)) * edge);
            sum += v.x; sum += v.y;
        }
        lo[i] = sum;
    }
}

extern "C" int rf_sumpool3d_2(const float* hi, int n, int c, int edge, float* lo, void* stream) {
    RF_REQUIRE(hi && lo && n > 0 && c > 0 && edge >= 2 && edge % 2 == 0, RF_E_INVALID, "rf_sumpool3d_2: bad arguments");
    const int half = edge / 2;
    const size_t cells = (size_t)n * c * half * half * half, want = (cells + 255) / 256;
    hipLaunchKernelGGL(k_sumpool2, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0, (hipStream_t)stream, hi, cells, half, lo);
    RF_CHECK_LAUNCH("rf_sumpool3d_2");
    return RF_OK;
}

// ------------------------------------------------------------------------------------------- GroupNorm backward
// Four launches, every tensor read once per pass and every launch wide enough for the chip whatever the shape ([4][16][64^3] of the final
// decoder = 64 (sample, channel) pairs; [1024][8][16^3] of the retrieval backbone = 8192):
//   pass 1  k_gnb_partial   per (sample, channel, slice of the volume): sum x, sum x^2, sum d, sum d * x  (d = d xn) in float64 -- raw sums, so
//                           the pass needs no moments and reads x and d exactly once (a product of two floats is exact in float64);
//   pass 2  k_gnb_finalize  per (sample, group): slices and channels summed in a fixed order -> mean, rstd, a1_c = sum d, a2_c = sum d * xh =
//                           rstd * (sum d x - mean * sum d), G1 / G2 = the group's gamma-weighted sums -> the coefficients of pass 3 and the
//                           per-sample pieces of dgamma / dbeta;
//   pass 3  k_gnb_apply     dx = k0 * d - k1 - xh * k2   (float64 per element, rounded once);
//   pass 4  k_gnb_sum_n     dgamma, dbeta = the per-sample pieces summed over the batch (float64, fixed order).
// (The first form recomputed the group moments in every channel's workgroup -- cpg re-reads of x -- with one workgroup per (sample, channel):
// 3.9 ms of a 26.7 ms training step for 0.5 ms of traffic.)
namespace {
inline int gnb_slices(size_t vol) {                               // >= 4096 elements per workgroup of pass 1, at most 64 slices per channel
    const size_t p = vol / 4096;
    return p < 1 ? 1 : (p > 64 ? 64 : (int)p);
}
}

// L = lanes per (sample, channel) row = 256 for volumes of >= 1024 elements, else the power of two >= the row's units (16-byte units; elements for
// 1^3), at least 1: a workgroup takes 256 / L rows (a 4^3 row is 16 units: one row per workgroup left 240 of 256 threads idle -- 0.42 ms for
// [1024][192][4^3]).  Sums: butterflies inside the row's lanes, then across its waves in order.
__global__ __launch_bounds__(256) void k_gnb_partial(const float* __restrict__ x, const float* __restrict__ dxn, size_t vol, int P, int L, int rows,
                                                     double4* __restrict__ part) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x * (256 / L) + tid / L, li = tid % L, p = blockIdx.y;
    __shared__ double red[4][4];
    double sx = 0.0, sq = 0.0, sd = 0.0, sdx = 0.0;
    if (row < rows) {
        const float* xc = x + (size_t)row * vol;
        const float* dc = dxn + (size_t)row * vol;
        if ((vol & 3) == 0) {                                       // 16-byte loads (every volume but 1^3)
            const size_t u = vol / 4, i0 = u * p / P, i1 = u * (p + 1) / P;
            const float4* xc4 = reinterpret_cast<const float4*>(xc);
            const float4* dc4 = reinterpret_cast<const float4*>(dc);
            for (size_t i = i0 + li; i < i1; i += L) {
                const float4 xv = xc4[i], dv = dc4[i];
                const double x0 = xv.x, x1 = xv.y, x2 = xv.z, x3 = xv.w, d0 = dv.x, d1 = dv.y, d2 = dv.z, d3 = dv.w;
                sx += (x0 + x1) + (x2 + x3);
                sq += (x0 * x0 + x1 * x1) + (x2 * x2 + x3 * x3);
                sd += (d0 + d1) + (d2 + d3);
                sdx += (d0 * x0 + d1 * x1) + (d2 * x2 + d3 * x3);
            }
        } else {
            const size_t i0 = vol * p / P, i1 = vol * (p + 1) / P;
            for (size_t i = i0 + li; i < i1; i += L) {
                const double xv = xc[i], dv = dc[i];
                sx += xv; sq += xv * xv; sd += dv; sdx += dv * xv;
            }
        }
    }
    const int span = L < 64 ? L : 64;
    for (int d = span >> 1; d >= 1; d >>= 1) {
        sx += __shfl_xor(sx, d, 64); sq += __shfl_xor(sq, d, 64); sd += __shfl_xor(sd, d, 64); sdx += __shfl_xor(sdx, d, 64);
    }
    if (L <= 64) {
        if (li == 0 && row < rows) part[(size_t)row * P + p] = make_double4(sx, sq, sd, sdx);
        return;
    }
    if (lane == 0) { red[wave][0] = sx; red[wave][1] = sq; red[wave][2] = sd; red[wave][3] = sdx; }
    __syncthreads();
    if (li == 0 && row < rows) {                                    // L = 128 (two rows per workgroup: waves 0-1, 2-3) or 256
        const int w0 = wave, nw = L / 64;
        double4 o = make_double4(0.0, 0.0, 0.0, 0.0);
        for (int w = w0; w < w0 + nw; ++w) { o.x += red[w][0]; o.y += red[w][1]; o.z += red[w][2]; o.w += red[w][3]; }
        part[(size_t)row * P + p] = o;
    }
}

// one 64-thread workgroup per (sample, group); thread k sums the slices of channel k (k, k + 64, ...) in slice order, thread 0 the channels in order
__global__ __launch_bounds__(64) void k_gnb_finalize(const double4* __restrict__ part, const float* __restrict__ gamma, int C, int cpg, size_t vol, int P,
                                                     double eps, const float* __restrict__ inv_scale, double2* __restrict__ moments, double4* __restrict__ coef,
                                                     double* __restrict__ a1_out, double* __restrict__ a2_out) {
    const int groups = C / cpg, nn = blockIdx.x / groups, g = blockIdx.x % groups, tid = threadIdx.x;
    __shared__ double4 ch[64];
    __shared__ double grp[4];                                       // mean, rstd, G1, G2
    const double m = (double)cpg * (double)vol;
    double S = 0.0, Q = 0.0;
    for (int k0 = 0; k0 < cpg; k0 += 64) {                          // moments first: the channels' a2 need the mean
        if (k0 + tid < cpg) {
            const double4* pp = part + ((size_t)nn * C + (size_t)g * cpg + k0 + tid) * P;
            double4 o = make_double4(0.0, 0.0, 0.0, 0.0);
            for (int p = 0; p < P; ++p) { const double4 v = pp[p]; o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w; }
            ch[tid] = o;
        }
        __syncthreads();
        if (tid == 0)
            for (int k = 0; k < 64 && k0 + k < cpg; ++k) { S += ch[k].x; Q += ch[k].y; }
        __syncthreads();
    }
    if (tid == 0) {
        const double mean = S / m;
        double var = Q / m - mean * mean;
        if (var < 0.0) var = 0.0;
        grp[0] = mean; grp[1] = 1.0 / sqrt(var + eps); grp[2] = 0.0; grp[3] = 0.0;
    }
    __syncthreads();
    const double mean = grp[0], rstd = grp[1];
    for (int k0 = 0; k0 < cpg; k0 += 64) {
        const int cc = g * cpg + k0 + tid;
        if (k0 + tid < cpg) {
            double4 o = ch[tid];
            if (cpg > 64) {                                         // more than one round: the sums were overwritten, take them again
                const double4* pp = part + ((size_t)nn * C + cc) * P;
                o = make_double4(0.0, 0.0, 0.0, 0.0);
                for (int p = 0; p < P; ++p) { const double4 v = pp[p]; o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w; }
            }
            const double a1 = o.z, a2 = rstd * (o.w - mean * o.z);
            a1_out[(size_t)nn * C + cc] = a1;
            a2_out[(size_t)nn * C + cc] = a2;
            ch[tid] = make_double4((double)gamma[cc] * a1, (double)gamma[cc] * a2, 0.0, 0.0);
        }
        __syncthreads();
        if (tid == 0)
            for (int k = 0; k < 64 && k0 + k < cpg; ++k) { grp[2] += ch[k].x; grp[3] += ch[k].y; }
        __syncthreads();
    }
    const double G1 = grp[2], G2 = grp[3];
    const double inv = inv_scale ? (double)*inv_scale : 1.0;       // d xn arrived multiplied by 1 / inv (a power of two): dx is linear in it
    for (int k = tid; k < cpg; k += 64) {
        const int cc = g * cpg + k;
        moments[(size_t)nn * C + cc] = make_double2(mean, rstd);
        coef[(size_t)nn * C + cc] = make_double4(inv * rstd * (double)gamma[cc], inv * rstd * G1 / m, inv * rstd * G2 / m, 0.0);
    }
}

// flat over the tensor's 16-byte units (elements for 1^3): a row's coefficients come through the cache, rows of any length fill the workgroups
__global__ __launch_bounds__(256) void k_gnb_apply(const float* __restrict__ x, const float* __restrict__ dxn, size_t vol, size_t rows, const double2* __restrict__ moments,
                                                   const double4* __restrict__ coef, float* __restrict__ dx) {
    if ((vol & 3) == 0) {
        const size_t u = vol / 4, total = rows * u;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const float4* d4 = reinterpret_cast<const float4*>(dxn);
        float4* o4 = reinterpret_cast<float4*>(dx);
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const size_t nc = i / u;
            const double2 mo = moments[nc];
            const double4 kf = coef[nc];
            const float4 xv = x4[i], dv = d4[i];
            float4 r;
            r.x = (float)(kf.x * (double)dv.x - kf.y - (((double)xv.x - mo.x) * mo.y) * kf.z);
            r.y = (float)(kf.x * (double)dv.y - kf.y - (((double)xv.y - mo.x) * mo.y) * kf.z);
            r.z = (float)(kf.x * (double)dv.z - kf.y - (((double)xv.z - mo.x) * mo.y) * kf.z);
            r.w = (float)(kf.x * (double)dv.w - kf.y - (((double)xv.w - mo.x) * mo.y) * kf.z);
            o4[i] = r;
        }
    } else {
        const size_t total = rows * vol;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const size_t nc = i / vol;
            const double2 mo = moments[nc];
            const double4 kf = coef[nc];
            const double xh = ((double)x[i] - mo.x) * mo.y;
            dx[i] = (float)(kf.x * (double)dxn[i] - kf.y - xh * kf.z);
        }
    }
}

// dgamma_c = sum_n a2[n][c], dbeta_c = sum_n a1[n][c] (x inv): 64 threads per channel, thread t sums samples t, t + 64, ... in float64, the 64 partial
// sums are added in thread order -- a fixed order
__global__ __launch_bounds__(64) void k_gnb_sum_n(const double* __restrict__ a1, const double* __restrict__ a2, int n, int C, const float* __restrict__ inv_scale,
                                                  float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x, t = threadIdx.x;
    __shared__ double r1[64], r2[64];
    double s1 = 0.0, s2 = 0.0;
    for (int i = t; i < n; i += 64) { s1 += a1[(size_t)i * C + c]; s2 += a2[(size_t)i * C + c]; }
    r1[t] = s1; r2[t] = s2;
    __syncthreads();
    if (t == 0) {
        double S1 = 0.0, S2 = 0.0;
#pragma unroll 4
        for (int k = 0; k < 64; ++k) { S1 += r1[k]; S2 += r2[k]; }
        const double inv = inv_scale ? (double)*inv_scale : 1.0;
        dbeta[c] = (float)(S1 * inv);
        dgamma[c] = (float)(S2 * inv);
    }
}

extern "C" size_t rf_gn_backward_ws_bytes(int n, int c, int edge) {
    const size_t vol = (size_t)edge * edge * edge;
    return (size_t)n * c * ((size_t)gnb_slices(vol) * sizeof(double4) + sizeof(double2) + sizeof(double4) + 2 * sizeof(double));
}

// x, dxn [n][c][edge^3]; gamma [c]; dxn_inv_scale: null, or a device float: d xn arrives multiplied by 1 / *dxn_inv_scale (the split data-gradient conv
// works on a power-of-two scaled dz, rf_dgrad_scale_affine) and everything that leaves is multiplied by it; out: dx [n][c][edge^3], dgamma [c], dbeta [c]
extern "C" int rf_gn_backward(const float* x, const float* dxn, int n, int c, int edge, const float* gamma, int groups, float eps, const float* dxn_inv_scale,
                              float* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream) {
    RF_REQUIRE(x && dxn && gamma && dx && dgamma && dbeta && ws && n > 0 && c > 0 && edge > 0 && groups > 0 && c % groups == 0, RF_E_INVALID,
               "rf_gn_backward: bad arguments");
    RF_REQUIRE(ws_bytes >= rf_gn_backward_ws_bytes(n, c, edge), RF_E_WORKSPACE, "rf_gn_backward: workspace too small");
    const size_t vol = (size_t)edge * edge * edge;
    const int cpg = c / groups, P = gnb_slices(vol);
    double4* part = (double4*)ws;                                    // [n * c][P]
    double4* coef = part + (size_t)n * c * P;                        // [n * c]
    double2* moments = (double2*)(coef + (size_t)n * c);             // [n * c]
    double* a1 = (double*)(moments + (size_t)n * c);                 // [n][c] sum d xn          (per-sample pieces of dbeta)
    double* a2 = a1 + (size_t)n * c;                                 // [n][c] sum d xn * xh     (... of dgamma)
    hipStream_t s = (hipStream_t)stream;
    const size_t units = (vol & 3) == 0 ? vol / 4 : vol;
    int L = 1;
    while (L < 256 && (size_t)L < units) L <<= 1;                   // lanes per row: the power of two >= its units, at most a workgroup
    const int rows = n * c, rpw = 256 / L;
    hipLaunchKernelGGL(k_gnb_partial, dim3((unsigned)((rows + rpw - 1) / rpw), P), dim3(256), 0, s, x, dxn, vol, P, L, rows, part);
    RF_CHECK_LAUNCH("rf_gn_backward(partial)");
    hipLaunchKernelGGL(k_gnb_finalize, dim3(n * groups), dim3(64), 0, s, part, gamma, c, cpg, vol, P, (double)eps, dxn_inv_scale, moments, coef, a1, a2);
    RF_CHECK_LAUNCH("rf_gn_backward(finalize)");
    const size_t want = ((size_t)rows * units + 255) / 256;
    hipLaunchKernelGGL(k_gnb_apply, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0, s, x, dxn, vol, (size_t)rows, moments, coef, dx);
    RF_CHECK_LAUNCH("rf_gn_backward(apply)");
    hipLaunchKernelGGL(k_gnb_sum_n, dim3(c), dim3(64), 0, s, a1, a2, n, c, dxn_inv_scale, dgamma, dbeta);
    RF_CHECK_LAUNCH("rf_gn_backward(sum)");
    return RF_OK;
}

// ------------------------------------------------------------------------------------------------- weight gradient
// dW[co][ci][tap] = sum over samples and voxels of dz[n][co][v] * xn[n][ci][v + tap - 1]   (xn = GroupNorm(x), zero padded)
// as an fp32-MFMA GEMM per workgroup:  D[co][col] += sum_k A[co][k] * B[k][col]  with k = voxels of an 8^3 box (4 per MFMA step),
// M = a 16-cout block, N = the 4 channels x 27 taps of one channel chunk (108 columns -> 7 n-blocks of 16, one per wave; wave 7
// helps staging only).  A = dz tile [16][512] in LDS, B = the GroupNorm-applied halo box [4][10^3] in LDS read through a per-lane
// column offset (channel * 1000 + tap offset).  A workgroup walks the boxes b = g, g + GB, ... of its (channel chunk, cout block)
// and keeps its 16 x 108 partial in registers; partials [GB][cout][cin][27] are then reduced in float64 in a fixed order.
// Edge >= 4 volumes (4^3: eight whole samples per box); 2^3 / 1^3 layers go through rf_linear_wgrad on the unfolded input (rfuse/autograd.py).
struct WgradArgs {
    const float* x;
    const float4* affine;
    const float* dz;
    float* parts;             // [GB][cout][cin][27]
    int cin, cout, n, edge, gb;
};

// S4: whole 4^3 samples, 8 per "box" (512 voxels = sample * 64 + voxel; halo image 8 x 6^3 per channel): the 4^3 levels of the retrieval
// backbone (192 -> 64, 64 -> 64, 32 -> 64 ... on 1024 samples) without the im2col detour (a [65536][5184] fp32 matrix written and read back).
template <bool S4>
__global__ __launch_bounds__(512) void k_conv3_wgrad(WgradArgs a) {
    constexpr int HE = S4 ? 6 : 10, CH = S4 ? 8 * 216 : 1000, DZS = 512 + 4;
    __shared__ float xs[4 * CH];
    __shared__ __attribute__((aligned(16))) float ds[16 * DZS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int cchunk = blockIdx.x, cob = blockIdx.y * 16, g = blockIdx.z;
    const int edge = a.edge, bpe = S4 ? 1 : edge / 8, boxes_per_sample = bpe * bpe * bpe;
    const int nboxes = S4 ? (a.n + 7) / 8 : a.n * boxes_per_sample;
    const size_t vol = (size_t)edge * edge * edge;
    // this lane's B column: n-block = wave, column c = wave*16 + li -> (channel, tap)
    const int col = wave * 16 + li;
    const bool col_ok = wave < 7 && col < 108;
    const int cch = col_ok ? col / 27 : 0, ctap = col_ok ? col % 27 : 0;
    const int coloff = cch * CH + ((ctap / 9) * HE + (ctap / 3) % 3) * HE + ctap % 3;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int b = g; b < nboxes; b += a.gb) {
        const int nn = S4 ? b * 8 : b / boxes_per_sample, bb = S4 ? 0 : b % boxes_per_sample;
        const int x0 = (bb % bpe) * 8, y0 = ((bb / bpe) % bpe) * 8, z0 = (bb / (bpe * bpe)) * 8;
        __syncthreads();                                            // previous box fully consumed
        for (int i = tid; i < 4 * CH; i += 512) {                   // GroupNorm-applied halo box, zero outside the volume / past cin
            const int c = i / CH, r = i % CH, ci = cchunk * 4 + c;
            float v = 0.f;
            if constexpr (S4) {
                const int sm = r / 216, q = r % 216;
                const int x = q % 6 - 1, y = (q / 6) % 6 - 1, z = q / 36 - 1;
                if (ci < a.cin && nn + sm < a.n && (unsigned)z < 4u && (unsigned)y < 4u && (unsigned)x < 4u) {
                    const float4 af = a.affine[(size_t)(nn + sm) * a.cin + ci];
                    v = fmaf(a.x[((size_t)(nn + sm) * a.cin + ci) * 64 + (z * 4 + y) * 4 + x] - af.x, af.y, af.z);
                }
            } else {
                const int hx = r % HE, hy = (r / HE) % HE, hz = r / (HE * HE);
                const int z = z0 + hz - 1, y = y0 + hy - 1, x = x0 + hx - 1;
                if (ci < a.cin && (unsigned)z < (unsigned)edge && (unsigned)y < (unsigned)edge && (unsigned)x < (unsigned)edge) {
                    const float4 af = a.affine[(size_t)nn * a.cin + ci];
                    v = fmaf(a.x[((size_t)nn * a.cin + ci) * vol + ((size_t)z * edge + y) * edge + x] - af.x, af.y, af.z);
                }
            }
            xs[i] = v;
        }
        for (int i = tid; i < 16 * 512; i += 512) {                 // dz tile [16 couts][512 voxels of the box]
            const int co = i >> 9, v = i & 511;
            float d = 0.f;
            if constexpr (S4) {
                if (cob + co < a.cout && nn + (v >> 6) < a.n) d = a.dz[((size_t)(nn + (v >> 6)) * a.cout + cob + co) * 64 + (v & 63)];
            } else {
                const int x = v & 7, y = (v >> 3) & 7, z = v >> 6;
                if (cob + co < a.cout) d = a.dz[((size_t)nn * a.cout + cob + co) * vol + ((size_t)(z0 + z) * edge + (y0 + y)) * edge + (x0 + x)];
            }
            ds[co * DZS + v] = d;
        }
        __syncthreads();
        if (wave < 7) {
#pragma unroll 4
            for (int v0 = 0; v0 < 512; v0 += 4) {
                const int v = v0 + kq;                              // this lane's voxel of the k-step
                const int hb = S4 ? (v >> 6) * 216 + (((v >> 4) & 3) * HE + ((v >> 2) & 3)) * HE + (v & 3)
                                  : (((v >> 6)) * HE + ((v >> 3) & 7)) * HE + (v & 7);
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
    const long long boxes = edge == 4 ? (n + 7) / 8 : (long long)n * (edge / 8) * (edge / 8) * (edge / 8);
    return (int)(boxes < 256 ? boxes : 256);         // workgroups per (channel chunk, cout block): 64 left most CUs with one 8-wave workgroup
}

extern "C" size_t rf_conv3d_k3_wgrad_ws_bytes(int cin, int cout, int n, int edge) {
    return (size_t)wgrad_groups(n, edge) * cout * cin * 27 * sizeof(float);
}

// x [n][cin][edge^3] (the layer input, single source), gn_affine as the forward, dz [n][cout][edge^3] -> dw OIDHW [cout][cin][27]
extern "C" int rf_conv3d_k3_wgrad(const float* x, int cin, int n, int edge, const float* gn_affine, const float* dz, int cout, float* dw, void* ws,
                                  size_t ws_bytes, void* stream) {
    RF_REQUIRE(x && gn_affine && dz && dw && ws && cin > 0 && cout > 0 && n > 0, RF_E_INVALID, "rf_conv3d_k3_wgrad: bad arguments");
    RF_REQUIRE(rf_is_pow2(edge) && edge >= 4 && edge <= 128, RF_E_UNSUPPORTED, "rf_conv3d_k3_wgrad: edge %d (8^3 boxes or whole 4^3 samples: a power of two >= 4)", edge);
    RF_REQUIRE(ws_bytes >= rf_conv3d_k3_wgrad_ws_bytes(cin, cout, n, edge), RF_E_WORKSPACE, "rf_conv3d_k3_wgrad: workspace too small");
    WgradArgs a;
    a.x = x; a.affine = reinterpret_cast<const float4*>(gn_affine); a.dz = dz; a.parts = (float*)ws;
    a.cin = cin; a.cout = cout; a.n = n; a.edge = edge; a.gb = wgrad_groups(n, edge);
    hipStream_t s = (hipStream_t)stream;
    if (edge == 4) hipLaunchKernelGGL(k_conv3_wgrad<true>, dim3((cin + 3) / 4, (cout + 15) / 16, a.gb), dim3(512), 0, s, a);
    else hipLaunchKernelGGL(k_conv3_wgrad<false>, dim3((cin + 3) / 4, (cout + 15) / 16, a.gb), dim3(512), 0, s, a);
    RF_CHECK_LAUNCH("rf_conv3d_k3_wgrad");
    const size_t count = (size_t)cout * cin * 27;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, (const float*)ws, a.gb, count, dw);
    RF_CHECK_LAUNCH("rf_conv3d_k3_wgrad(reduce)");
    return RF_OK;
}
