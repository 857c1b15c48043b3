// 3D U-Net primitives for gfx950: GroupNorm statistics, 3x3x3 convolution as an fp32-MFMA implicit GEMM with the
// GroupNorm apply, nearest-upsample + concat read and ReLU fused in, max-pool, and the final 1x1x1 conv + tanh.
//
// Reference arithmetic being replaced: model/unet.py:19-76 (SingleConv 'gcr'), :237 (MaxPool3d 2), :297-308 and
// :354-360 (upsample + concat), model/refinement.py:54-55 (1x1x1 conv + tanh).
//
// The MFMA convolution itself lives in conv3d_mfma.hip.  Original design notes (rf_conv3d_k3_gn_relu)
//   GEMM view: M = output voxels, N = cout, K = cin*27.  One workgroup (4 waves) owns P = 512 output voxels -- an
//   8^3 box of one sample, or 8 whole 4^3 samples, or 64 whole 2^3 samples -- and up to 64 output channels.
//   K is walked in chunks of 4 input channels: the chunk's halo box (GroupNorm already applied, zero outside the
//   volume, upsample/concat resolved) and its [27][4][cout] weight slab are staged in LDS; then for each of the 27
//   taps one v_mfma_f32_16x16x4_f32 k-step multiplies 16 voxels x 4 channels by 4 channels x 16 couts.  A operands
//   are im2col reads straight out of the halo box (lane -> (voxel = lane&15, channel = lane>>4)), so the tap offset
//   is a compile-time LDS immediate; B operands are rows of the weight slab.  Each wave keeps MB x NB accumulator
//   tiles (8 x 4 x 4 VGPRs) so one A read is reused by NB MFMAs and one B read by MB MFMAs.
//   fp32-input MFMA is bit-for-bit an fp32 FMA chain in k order (CDNA4 guide), so this is exact fp32 arithmetic.
#include "common.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------ weight pack
__global__ void k_conv3_pack(const float* __restrict__ w, int cout, int cin, int cin4, int cout16, float* __restrict__ wp) {
    const size_t total = (size_t)27 * cin4 * cout16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % cout16);
        const int ci = (int)((i / cout16) % cin4);
        const int tap = (int)(i / ((size_t)cout16 * cin4));
        float v = 0.f;
        if (co < cout && ci < cin) v = w[((size_t)co * cin + ci) * 27 + tap];
        wp[i] = v;
    }
}

extern "C" size_t rf_conv3_packed_floats(int cout, int cin) {
    return (size_t)27 * rf_round_up(cin, 4) * rf_round_up(cout, 16);
}

extern "C" int rf_conv3_pack_weight(const float* w, int cout, int cin, float* wp, void* stream) {
    RF_REQUIRE(w && wp && cout > 0 && cin > 0, RF_E_INVALID, "rf_conv3_pack_weight: bad arguments");
    const size_t total = rf_conv3_packed_floats(cout, cin);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_conv3_pack, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, cout, cin,
                       rf_round_up(cin, 4), rf_round_up(cout, 16), wp);
    RF_CHECK_LAUNCH("rf_conv3_pack_weight");
    return RF_OK;
}

// ------------------------------------------------------------------------------------------- GroupNorm stats
// pass 1: per (sample, group, slice) partial sum / sum of squares in float64; upsampled channels weigh 8x.
__global__ __launch_bounds__(256) void k_gn_partial(const float* __restrict__ src0, int c0, const float* __restrict__ src1, int c1,
                                                    size_t vol0, size_t vol1, int groups, int cpg, int slices,
                                                    double2* __restrict__ part) {
    const int ng = blockIdx.x, s = blockIdx.y;
    const int nn = ng / groups, g = ng % groups;
    const int tid = threadIdx.x;
    double sum = 0.0, sq = 0.0;
    // the group's channels are contiguous in memory within each source: at most two segments (src0 part, src1 part)
    const int ca = g * cpg, cb = ca + cpg;
    for (int seg = 0; seg < 2; ++seg) {
        const float* base;
        size_t len;
        double wgt;
        if (seg == 0) {
            const int hi_c = cb < c0 ? cb : c0;
            if (ca >= hi_c) continue;
            base = src0 + ((size_t)nn * c0 + ca) * vol0; len = (size_t)(hi_c - ca) * vol0; wgt = 1.0;
        } else {
            const int lo_c = ca > c0 ? ca : c0;
            if (lo_c >= cb) continue;
            base = src1 + ((size_t)nn * c1 + (lo_c - c0)) * vol1; len = (size_t)(cb - lo_c) * vol1; wgt = 8.0;
        }
        double ls = 0.0, lq = 0.0;
        if ((len & 3) == 0 && (reinterpret_cast<uintptr_t>(base) & 15) == 0) {
            const size_t len4 = len >> 2;
            const size_t lo = len4 * s / slices, hi = len4 * (s + 1) / slices;
            const float4* b4 = reinterpret_cast<const float4*>(base);
            for (size_t i = lo + tid; i < hi; i += 256) {
                const float4 v = b4[i];
                ls += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
                lq += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
            }
        } else {
            const size_t lo = len * s / slices, hi = len * (s + 1) / slices;
            for (size_t i = lo + tid; i < hi; i += 256) {
                const float v = base[i];
                ls += v;
                lq += (double)v * v;
            }
        }
        sum += wgt * ls;
        sq += wgt * lq;
    }
    sum = wave_sum(sum);
    sq = wave_sum(sq);
    __shared__ double red[8];
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) { red[wave * 2] = sum; red[wave * 2 + 1] = sq; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < 4; ++w) { a += red[w * 2]; b += red[w * 2 + 1]; }
        part[(size_t)ng * slices + s] = make_double2(a, b);
    }
}

// pass 2: fixed-order reduction of the slices, then the affine triple per (n, c)
__global__ void k_gn_finalize(const double2* __restrict__ part, int n, int C, int groups, int cpg, int slices, double count,
                              const float* __restrict__ gamma, const float* __restrict__ beta, double eps,
                              float4* __restrict__ affine) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * C) return;
    const int nn = i / C, c = i % C, g = c / cpg;
    double a = 0.0, b = 0.0;
    for (int s = 0; s < slices; ++s) {
        const double2 p = part[((size_t)nn * groups + g) * slices + s];
        a += p.x;
        b += p.y;
    }
    const double mean = a / count;
    double var = b / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + eps);
    affine[i] = gn_affine(mean, rstd, gamma[c], beta[c]);
}

static int gn_slices(size_t group_elems) {
    size_t s = (group_elems + 16383) / 16384;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return (int)s;
}

extern "C" size_t rf_gn_stats_ws_bytes(int n, int groups) { return (size_t)n * groups * 64 * sizeof(double2); }

extern "C" int rf_gn_stats(const float* src0, int c0, const float* src1, int c1, int n, int edge,
                           const float* gamma, const float* beta, int groups, float eps,
                           float* gn_affine, void* ws, size_t ws_bytes, void* stream) {
    const int C = c0 + c1;
    RF_REQUIRE(n > 0 && C > 0 && groups > 0 && C % groups == 0, RF_E_INVALID, "rf_gn_stats: channels %d not divisible by groups %d", C, groups);
    RF_REQUIRE(rf_is_pow2(edge) && edge <= 128, RF_E_INVALID, "rf_gn_stats: edge %d must be a power of two <= 128", edge);
    RF_REQUIRE((c0 == 0 || src0) && (c1 == 0 || src1) && gamma && beta && gn_affine && ws, RF_E_INVALID, "rf_gn_stats: null pointer");
    RF_REQUIRE(c1 == 0 || edge >= 2, RF_E_INVALID, "rf_gn_stats: upsampled source needs edge >= 2");
    RF_REQUIRE(ws_bytes >= rf_gn_stats_ws_bytes(n, groups), RF_E_WORKSPACE, "rf_gn_stats: workspace too small");
    const size_t vol0 = (size_t)edge * edge * edge, vol1 = vol0 / 8;
    const int cpg = C / groups;
    const int slices = gn_slices((size_t)cpg * vol0);
    hipLaunchKernelGGL(k_gn_partial, dim3(n * groups, slices), dim3(256), 0, (hipStream_t)stream, src0, c0, src1, c1, vol0, vol1,
                       groups, cpg, slices, (double2*)ws);
    RF_CHECK_LAUNCH("rf_gn_stats(partial)");
    hipLaunchKernelGGL(k_gn_finalize, dim3((n * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const double2*)ws, n, C, groups, cpg,
                       slices, (double)cpg * (double)vol0, gamma, beta, (double)eps, reinterpret_cast<float4*>(gn_affine));
    RF_CHECK_LAUNCH("rf_gn_stats(finalize)");
    return RF_OK;
}

static int conv_check(const char* who, const float* src0, int c0, const float* src1, int c1, int n, int edge, const float* gn_affine,
                      const float* w, int cout, float* out) {
    RF_REQUIRE(n > 0 && c0 >= 0 && c1 >= 0 && c0 + c1 > 0 && cout > 0, RF_E_INVALID, "%s: bad sizes", who);
    RF_REQUIRE(rf_is_pow2(edge) && edge <= 128, RF_E_INVALID, "%s: edge %d must be a power of two <= 128", who, edge);
    RF_REQUIRE((c0 == 0 || src0) && (c1 == 0 || src1) && gn_affine && w && out, RF_E_INVALID, "%s: null pointer", who);
    RF_REQUIRE(c1 == 0 || edge >= 2, RF_E_INVALID, "%s: upsampled source needs edge >= 2", who);
    return RF_OK;
}

// ----------------------------------------------------------------------------------------------- conv, direct
// One thread per output element; plain fp32 FMAs in (ci, tap) order.  Cross-check path and the 1^3 path.
__global__ __launch_bounds__(256) void k_conv3_direct(const float* __restrict__ src0, int c0, const float* __restrict__ src1, int c1, int n,
                                                      int edge, const float4* __restrict__ affine,
                                                      const float* __restrict__ w, int cout, float* __restrict__ out) {
    const size_t vol = (size_t)edge * edge * edge;
    const size_t total = (size_t)n * cout * vol;
    const int cin = c0 + c1, half = edge >> 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % edge), y = (int)((i / edge) % edge), z = (int)((i / ((size_t)edge * edge)) % edge);
        const int co = (int)((i / vol) % cout), nn = (int)(i / (vol * cout));
        float acc = 0.f;
        for (int ci = 0; ci < cin; ++ci) {
            const float4 af = affine[(size_t)nn * cin + ci];
            const float* wk = w + ((size_t)co * cin + ci) * 27;
            for (int dz = 0; dz < 3; ++dz) {
                const int zz = z + dz - 1;
                if ((unsigned)zz >= (unsigned)edge) continue;
                for (int dy = 0; dy < 3; ++dy) {
                    const int yy = y + dy - 1;
                    if ((unsigned)yy >= (unsigned)edge) continue;
                    for (int dx = 0; dx < 3; ++dx) {
                        const int xx = x + dx - 1;
                        if ((unsigned)xx >= (unsigned)edge) continue;
                        float r;
                        if (ci < c0) r = src0[(((size_t)nn * c0 + ci) * edge + zz) * edge * edge + (size_t)yy * edge + xx];
                        else r = src1[(((size_t)nn * c1 + (ci - c0)) * half + (zz >> 1)) * half * half + (size_t)(yy >> 1) * half + (xx >> 1)];
                        acc = fmaf(fmaf(r - af.x, af.y, af.z), wk[(dz * 3 + dy) * 3 + dx], acc);
                    }
                }
            }
        }
        out[i] = fmaxf(acc, 0.f);
    }
}

extern "C" int rf_conv3d_k3_gn_relu_direct(const float* src0, int c0, const float* src1, int c1, int n, int edge,
                                           const float* gn_affine, const float* w_oidhw, int cout,
                                           float* out, void* stream) {
    int rc = conv_check("rf_conv3d_k3_gn_relu_direct", src0, c0, src1, c1, n, edge, gn_affine, w_oidhw, cout, out);
    if (rc) return rc;
    const size_t total = (size_t)n * cout * edge * edge * edge;
    const size_t want = (total + 255) / 256;
    const int blocks = (int)(want < 8192 ? want : 8192);
    hipLaunchKernelGGL(k_conv3_direct, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src0, c0, src1, c1, n, edge, reinterpret_cast<const float4*>(gn_affine),
                       w_oidhw, cout, out);
    RF_CHECK_LAUNCH("rf_conv3d_k3_gn_relu_direct");
    return RF_OK;
}

// --------------------------------------------------------------------------------------------------- max pool
// One workgroup per (plane = sample*channel, chunk of up to 2048 outputs); optionally emits the chunk's (sum, sum of
// squares) so the next GroupNorm never re-reads the pooled tensor.
#define RF_POOL_CHUNK 2048
// one WAVE per (plane, chunk): no LDS, no block barrier; 4 units per workgroup
__global__ __launch_bounds__(256) void k_maxpool2(const float* __restrict__ x, size_t units, int edge, int chunks, float* __restrict__ out,
                                                  double2* __restrict__ stats) {
    const int h = edge >> 1;
    const size_t ovol = (size_t)h * h * h;
    const size_t unit = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= units) return;
    const int lane = threadIdx.x & 63;
    const size_t plane = unit / chunks;
    const int chunk = (int)(unit % chunks);
    const float* src = x + plane * (size_t)edge * edge * edge;
    float* dst = out + plane * ovol;
    const size_t lo = (size_t)chunk * RF_POOL_CHUNK;
    const size_t hi = lo + RF_POOL_CHUNK < ovol ? lo + RF_POOL_CHUNK : ovol;
    double sm = 0.0, sq = 0.0;
    for (size_t i = lo + lane; i < hi; i += 64) {
        const int ox = (int)(i % h), oy = (int)((i / h) % h), oz = (int)(i / ((size_t)h * h));
        const float* b = src + ((size_t)(2 * oz) * edge + 2 * oy) * edge + 2 * ox;
        float m = -INFINITY;
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const float2 v = *reinterpret_cast<const float2*>(b + ((size_t)dz * edge + dy) * edge);
                m = fmaxf(m, fmaxf(v.x, v.y));
            }
        dst[i] = m;
        sm += (double)m;
        sq += (double)m * m;
    }
    if (stats) {
        sm = wave_sum(sm);
        sq = wave_sum(sq);
        if (lane == 0) stats[unit] = make_double2(sm, sq);
    }
}

// Small volumes (edge 2, 4, 8: 1, 8, 64 pooled voxels per plane -- the deep levels of the retrieval backbone, thousands of samples): a wave per
// plane leaves 63 / 56 / 0 lanes idle and pays one wave's launch + a 64-lane float64 reduction per 8 outputs (173 us for 64 ch x 8192 samples
// @4^3).  Here a wave takes 64 / OV planes, lane = (plane, pooled voxel); the statistics are a butterfly over the OV lanes of a plane.
template <int OV>
__global__ __launch_bounds__(256) void k_maxpool2_small(const float* __restrict__ x, size_t planes, float* __restrict__ out, double2* __restrict__ stats) {
    constexpr int H = OV == 1 ? 1 : (OV == 8 ? 2 : 4), E = 2 * H, PPW = 64 / OV;
    const size_t gl = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t plane = gl / OV;
    const int o = (int)(gl % OV);
    const bool live = plane < planes;
    float m = 0.f;
    if (live) {
        const int ox = o % H, oy = (o / H) % H, oz = o / (H * H);
        const float* b = x + plane * (size_t)(E * E * E) + ((size_t)(2 * oz) * E + 2 * oy) * E + 2 * ox;
        m = -INFINITY;
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const float2 v = *reinterpret_cast<const float2*>(b + ((size_t)dz * E + dy) * E);
                m = fmaxf(m, fmaxf(v.x, v.y));
            }
        out[plane * OV + o] = m;
    }
    if (stats) {
        double sm = (double)m, sq = (double)m * m;
#pragma unroll
        for (int d = OV >> 1; d > 0; d >>= 1) { sm += __shfl_xor(sm, d, 64); sq += __shfl_xor(sq, d, 64); }
        if (live && o == 0) stats[plane] = make_double2(sm, sq);
    }
    (void)PPW;
}

extern "C" int rf_maxpool_stats_tiles(int edge) {
    const size_t ovol = (size_t)(edge / 2) * (edge / 2) * (edge / 2);
    return (int)((ovol + RF_POOL_CHUNK - 1) / RF_POOL_CHUNK);
}

static int maxpool_impl(const float* x, int n, int c, int edge, float* out, double* stats, void* stream) {
    RF_REQUIRE(x && out && n > 0 && c > 0, RF_E_INVALID, "rf_maxpool3d_2: bad arguments");
    RF_REQUIRE(rf_is_pow2(edge) && edge >= 2 && edge <= 128, RF_E_INVALID, "rf_maxpool3d_2: edge %d", edge);
    if (edge <= 8) {
        const size_t planes = (size_t)n * c, ov = (size_t)(edge / 2) * (edge / 2) * (edge / 2);
        const unsigned blocks = (unsigned)((planes * ov + 255) / 256);
        double2* st = reinterpret_cast<double2*>(stats);
        if (edge == 2) hipLaunchKernelGGL(k_maxpool2_small<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, planes, out, st);
        else if (edge == 4) hipLaunchKernelGGL(k_maxpool2_small<8>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, planes, out, st);
        else hipLaunchKernelGGL(k_maxpool2_small<64>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, planes, out, st);
        RF_CHECK_LAUNCH("rf_maxpool3d_2");
        return RF_OK;
    }
    const int chunks = rf_maxpool_stats_tiles(edge);
    const size_t units = (size_t)n * c * chunks;
    hipLaunchKernelGGL(k_maxpool2, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, units, edge, chunks, out,
                       reinterpret_cast<double2*>(stats));
    RF_CHECK_LAUNCH("rf_maxpool3d_2");
    return RF_OK;
}

extern "C" int rf_maxpool3d_2(const float* x, int n, int c, int edge, float* out, void* stream) {
    return maxpool_impl(x, n, c, edge, out, nullptr, stream);
}

extern "C" int rf_maxpool3d_2_stats(const float* x, int n, int c, int edge, float* out, double* stats, void* stream) {
    RF_REQUIRE(stats, RF_E_INVALID, "rf_maxpool3d_2_stats: null stats buffer");
    return maxpool_impl(x, n, c, edge, out, stats, stream);
}

// ------------------------------------------------------------------------------- GroupNorm from fused statistics
// scale/shift of GroupNorm(cat(src0, up2(src1))) from per-(sample, channel, tile) partial sums emitted by the producers
// of src0 / src1 (rf_conv3d_k3_gn_relu_stats, rf_maxpool3d_2_stats).  One wave per (sample, group); fixed summation order.
// LPU = lanes per (sample, group) unit: 64 (one wave per unit) when a group spans many stats entries (big volumes, many
// tiles), 1 (one thread per unit) when it spans a handful (thousands of small samples) -- same fixed summation order.
template <int LPU>
__global__ __launch_bounds__(256) void k_gn_from_stats(const double2* __restrict__ st0, int c0, int t0, const double2* __restrict__ st1, int c1,
                                                       int t1, int units, int groups, int cpg, double count, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, double eps, float4* __restrict__ affine) {
    const int gtid = blockIdx.x * 256 + threadIdx.x;
    const int unit = gtid / LPU, lane = gtid % LPU;
    if (unit >= units) return;
    const int nn = unit / groups, g = unit % groups;
    const int C = c0 + c1, ca = g * cpg, cb = ca + cpg;
    double sm = 0.0, sq = 0.0;
    {   // src0 channels of the group: contiguous run of (channels x tiles) entries
        const int hi_c = cb < c0 ? cb : c0;
        if (ca < hi_c) {
            const double2* p = st0 + ((size_t)nn * c0 + ca) * t0;
            const int len = (hi_c - ca) * t0;
            for (int i = lane; i < len; i += LPU) { sm += p[i].x; sq += p[i].y; }
        }
    }
    {   // src1 (upsampled) channels: every low-res voxel is seen 8 times
        const int lo_c = ca > c0 ? ca : c0;
        if (lo_c < cb) {
            const double2* p = st1 + ((size_t)nn * c1 + (lo_c - c0)) * t1;
            const int len = (cb - lo_c) * t1;
            double a = 0.0, b = 0.0;
            for (int i = lane; i < len; i += LPU) { a += p[i].x; b += p[i].y; }
            sm += 8.0 * a;
            sq += 8.0 * b;
        }
    }
    if (LPU == 64) {
        sm = wave_sum(sm);
        sq = wave_sum(sq);
    }
    const double mean = sm / count;
    double var = sq / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + eps);
    for (int c = ca + lane; c < cb; c += LPU) affine[(size_t)nn * C + c] = gn_affine(mean, rstd, gamma[c], beta[c]);
}

extern "C" int rf_gn_from_stats(const double* stats0, int c0, int tiles0, const double* stats1, int c1, int tiles1, int n, int edge,
                                const float* gamma, const float* beta, int groups, float eps, float* gn_affine, void* stream) {
    const int C = c0 + c1;
    RF_REQUIRE(n > 0 && C > 0 && groups > 0 && C % groups == 0, RF_E_INVALID, "rf_gn_from_stats: channels %d not divisible by groups %d", C, groups);
    RF_REQUIRE((c0 == 0 || (stats0 && tiles0 > 0)) && (c1 == 0 || (stats1 && tiles1 > 0)) && gamma && beta && gn_affine, RF_E_INVALID,
               "rf_gn_from_stats: null pointer / zero tiles");
    RF_REQUIRE(rf_is_pow2(edge) && edge <= 128, RF_E_INVALID, "rf_gn_from_stats: edge %d", edge);
    const int cpg = C / groups;
    const int units = n * groups;
    const long long entries = (long long)cpg * (tiles0 > tiles1 ? tiles0 : tiles1);     // stats entries one unit sums (upper bound)
    if (entries <= 128 && units >= 1024)
        hipLaunchKernelGGL(k_gn_from_stats<1>, dim3((units + 255) / 256), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const double2*>(stats0), c0,
                           tiles0, reinterpret_cast<const double2*>(stats1), c1, tiles1, units, groups, cpg, (double)cpg * edge * edge * edge, gamma,
                           beta, (double)eps, reinterpret_cast<float4*>(gn_affine));
    else
        hipLaunchKernelGGL(k_gn_from_stats<64>, dim3((units + 3) / 4), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const double2*>(stats0), c0,
                           tiles0, reinterpret_cast<const double2*>(stats1), c1, tiles1, units, groups, cpg, (double)cpg * edge * edge * edge, gamma,
                           beta, (double)eps, reinterpret_cast<float4*>(gn_affine));
    RF_CHECK_LAUNCH("rf_gn_from_stats");
    return RF_OK;
}

// ------------------------------------------------------------------------------------------ 1x1x1 conv + tanh
__global__ __launch_bounds__(256) void k_conv1x1_tanh(const float* __restrict__ x, int n, int c, size_t vox, const float* __restrict__ w,
                                                      const float* __restrict__ b, float post_add, float post_mul, float* __restrict__ out) {
    const size_t vox4 = vox >> 2, total = (size_t)n * vox4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t nn = i / vox4, v4 = i % vox4;
        const float bias = b[0];
        float4 acc = make_float4(bias, bias, bias, bias);
        const float4* xp = reinterpret_cast<const float4*>(x + nn * c * vox) + v4;
        for (int ci = 0; ci < c; ++ci) {
            const float4 v = xp[(size_t)ci * vox4];
            const float wc = w[ci];
            acc.x = fmaf(v.x, wc, acc.x); acc.y = fmaf(v.y, wc, acc.y); acc.z = fmaf(v.z, wc, acc.z); acc.w = fmaf(v.w, wc, acc.w);
        }
        float4 o;
        o.x = (tanhf(acc.x) + post_add) * post_mul; o.y = (tanhf(acc.y) + post_add) * post_mul;
        o.z = (tanhf(acc.z) + post_add) * post_mul; o.w = (tanhf(acc.w) + post_add) * post_mul;
        reinterpret_cast<float4*>(out + nn * vox)[v4] = o;
    }
}

extern "C" int rf_conv1x1_tanh(const float* x, int n, int c, size_t voxels, const float* w, const float* b,
                               float post_add, float post_mul, float* out, void* stream) {
    RF_REQUIRE(x && w && b && out && n > 0 && c > 0 && voxels > 0, RF_E_INVALID, "rf_conv1x1_tanh: bad arguments");
    RF_REQUIRE((voxels & 3) == 0, RF_E_UNSUPPORTED, "rf_conv1x1_tanh: voxel count must be a multiple of 4");
    const size_t want = ((size_t)n * (voxels / 4) + 255) / 256;
    hipLaunchKernelGGL(k_conv1x1_tanh, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream, x, n, c, voxels, w, b,
                       post_add, post_mul, out);
    RF_CHECK_LAUNCH("rf_conv1x1_tanh");
    return RF_OK;
}

// ------------------------------------------------------------------------------- valid strided conv + LeakyReLU
// Direct form for the patch encoders (tiny batches of small windows).  One thread per output voxel and cout.
__global__ __launch_bounds__(256) void k_conv3_valid(const float* __restrict__ x, int n, int cin, int s, const float* __restrict__ w,
                                                     const float* __restrict__ bias, int cout, int k, int stride, float slope, int so,
                                                     float* __restrict__ out) {
    const size_t ovol = (size_t)so * so * so, total = (size_t)n * cout * ovol;
    const size_t ivol = (size_t)s * s * s;
    const int k3 = k * k * k;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % so), oy = (int)((i / so) % so), oz = (int)((i / ((size_t)so * so)) % so);
        const int co = (int)((i / ovol) % cout);
        const size_t nn = i / (ovol * cout);
        float acc = bias ? bias[co] : 0.f;
        const float* xb = x + nn * cin * ivol + ((size_t)(oz * stride) * s + oy * stride) * s + ox * stride;
        const float* wb = w + (size_t)co * cin * k3;
        for (int ci = 0; ci < cin; ++ci) {
            const float* xc = xb + ci * ivol;
            const float* wc = wb + ci * k3;
            for (int dz = 0; dz < k; ++dz)
                for (int dy = 0; dy < k; ++dy)
                    for (int dx = 0; dx < k; ++dx)
                        acc = fmaf(xc[((size_t)dz * s + dy) * s + dx], wc[(dz * k + dy) * k + dx], acc);
        }
        out[i] = acc > 0.f ? acc : acc * slope;
    }
}

extern "C" int rf_conv3d_valid_leaky(const float* x, int n, int cin, int s, const float* w, const float* bias, int cout, int k,
                                     int stride, float slope, float* out, void* stream) {
    RF_REQUIRE(x && w && out && n > 0 && cin > 0 && cout > 0 && k > 0 && stride > 0 && s >= k, RF_E_INVALID, "rf_conv3d_valid_leaky: bad arguments");
    const int so = (s - k) / stride + 1;
    const size_t total = (size_t)n * cout * so * so * so;
    const size_t want = (total + 255) / 256;
    hipLaunchKernelGGL(k_conv3_valid, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, (hipStream_t)stream, x, n, cin, s, w, bias,
                       cout, k, stride, slope, so, out);
    RF_CHECK_LAUNCH("rf_conv3d_valid_leaky");
    return RF_OK;
}
