// rf_conv3d_k3_wgrad_split: the weight gradient of SingleConv 'gcr' (reference model/unet.py:19-76; trained by trainer/train_refinement.py:41-43,
// 108-116) on the F16 matrix cores by OPERAND SPLITTING -- the arithmetic of conv3d_split.hip (x = h + l / 2^11 as two f16 values, exact f16 x f16
// products, fp32 accumulation in separate hi / lo accumulators) applied to
//
//     dW[co][ci][tap] = sum over samples and voxels of dz[n][co][v] * xn[n][ci][v + tap - 1]        (xn = GroupNorm(x), zero padded)
//
// as a GEMM  D[co][(ci, tap)] += sum_k A[co][k] * B[k][(ci, tap)]  with k = voxels.  rf_conv3d_k3_wgrad (conv3d_backward.hip) runs it with one
// v_mfma_f32_16x16x4_f32 per two scalar LDS reads and is bound by the fp32 matrix pipe (2.6 ms for 96 -> 56 @8^3 x 1024: 58 TFLOP/s); here a
// k-step is 32 voxels = FOUR X-ROWS of an 8^3 box (lane group g = lane >> 4 supplies row 4s + g, its 8 halves are the row's 8 voxels), so
//   A (dz)  = one 16-byte slot per (cout, row): the row's 8 gradients, scaled by the power of two of rf_dgrad_scale_affine (a gradient can be
//             anywhere; the split forms carry f16 pairs) and split -- LDS image [cout][row], staged once per half box;
//   B (xn)  = one 16-byte slot per (channel, halo row (z, y), x-shift): the row's voxels x0 + t - 1 .. x0 + t + 6 for the three tap columns
//             t = 0, 1, 2 -- the x-shift of a tap cannot be an address offset inside a 16-byte operand, so the staging thread of a row (it holds the
//             row's 10 values) writes the three shifted operands; the (dz, dy) part of a tap IS an address offset (another halo row).
// A workgroup (8 waves = 2 m-pairs x 4 n-groups) owns 64 couts x 8 input channels (216 columns = 13.5 n-blocks) and walks the boxes b = g, g + GB, ...
// in z-halves (LDS: 4 z planes of dz rows for 64 couts 66 KB, 6 halo planes x 10 rows x 3 shifts for 8 channels 45 KB); a wave holds a 2 x 4 (2 x 3)
// block of 16 x 16 tiles, hi and lo: 12 (10) operand reads per 24 (18) MFMAs.  Partials [GB][cout][cin][27] in fp32 (hi + lo / 2^11), reduced in float64 in a
// fixed order and rescaled (x 16 / s) by k_wgrad_split_reduce.  Edge >= 8 volumes, cin >= 6; the rest stays with rf_conv3d_k3_wgrad.
#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int WS_B_SLOTS = 1440;                            // per plane: 8 channels x 60 halo rows x 3 shifts (4^3: 4 samples x 4 channels x 30 row pairs x 3)
constexpr int WS_A_STRIDE = 33;                             // slots per cout: 32 rows + 1 (bank spread)
constexpr int WS_A_SLOTS = 64 * WS_A_STRIDE;                // 2112 per plane
constexpr int WS_LDS_BYTES = (WS_B_SLOTS + WS_A_SLOTS) * 2 * 16;       // 113,664
constexpr float WS_ACT_SCALE = 1.0f / 16, WS_LO = 2048.0f;
constexpr int ws_nc(bool s4) { return s4 ? 4 : 8; }         // input channels per workgroup
}

struct WgradSplitArgs {
    const float* x;
    const float4* affine;
    const float* dz;
    const float* scales;      // (s, 1 / s) of rf_dgrad_scale_affine
    float* parts;             // [GB][cout][cin][27]
    int cin, cout, n, edge, gb;
};

__device__ __forceinline__ void ws_split(float v, _Float16& h, _Float16& l) {
    v = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
    h = (_Float16)v;
    l = (_Float16)fmaf(-WS_LO, (float)h, v * WS_LO);
}

// S4: whole 4^3 samples, eight per "box", four per half.  A slot is two x-rows of a sample (z, y = 2 yp, 2 yp + 1; 8 contiguous floats of dz); the B image
// holds, per (sample, channel), the row PAIRS starting at every halo row hy0 = 0 .. 4 of every halo plane, x-shifted three ways (the x halo of a whole
// sample is always zero padding): [sample 4][channel 4][hz 6][hy0 5][shift 3] = the same 1440 slots; 108 columns = 7 n-blocks over the 4 n-groups.
template <bool S4>
__global__ __launch_bounds__(512, 2) void k_conv3_wgrad_split(WgradSplitArgs a) {
    constexpr int NC = ws_nc(S4), NT = S4 ? 7 : 14, JM = S4 ? 2 : 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    h8* Bh = reinterpret_cast<h8*>(lds_raw);
    h8* Bl = Bh + WS_B_SLOTS;
    h8* Ah = Bl + WS_B_SLOTS;
    h8* Al = Ah + WS_A_SLOTS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 15, kg = lane >> 4;
    const int cchunk = blockIdx.x, cob = blockIdx.y * 64, g = blockIdx.z;
    const int edge = a.edge, bpe = S4 ? 1 : edge / 8, bps = bpe * bpe * bpe;
    const int nboxes = S4 ? (a.n + 7) / 8 : a.n * bps;
    const size_t vol = (size_t)edge * edge * edge;
    const int cin = a.cin, cout = a.cout;

    // this lane's B columns: n-block nt = wn + 4 j, column = nt * 16 + li -> (channel, tap); the k-step adds a compile-time row offset
    int bbase[JM];
#pragma unroll
    for (int j = 0; j < JM; ++j) {
        const int col = (wn + 4 * j) * 16 + li;
        const bool ok = col < NC * 27;
        const int ci = ok ? col / 27 : 0, tap = ok ? col % 27 : 0;
        bbase[j] = S4 ? ((ci * 6 + tap / 9 + (kg >> 1)) * 5 + (tap / 3) % 3 + 2 * (kg & 1)) * 3 + tap % 3
                      : ((ci * 6 + tap / 9) * 10 + (tap / 3) % 3 + kg) * 3 + tap % 3;
    }
    const int nj = (NT - wn + 3) / 4;                               // n-blocks wn, wn + 4, ... < NT
    int abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) abase[i] = ((2 * wm + i) * 16 + li) * WS_A_STRIDE + kg;
    const bool m_live = cob + 32 * wm < cout;                       // the wave's 32 couts exist (wave-uniform)

    f32x4 hi[2][JM], lo[2][JM];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < JM; ++j) { hi[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const float sdz = a.scales[0];

    for (int b = g; b < nboxes; b += a.gb) {
        const int nn = S4 ? b * 8 : b / bps, bb = S4 ? 0 : b % bps;
        const int x0 = (bb % bpe) * 8, y0 = ((bb / bpe) % bpe) * 8, z0 = (bb / (bpe * bpe)) * 8;
        for (int zh = 0; zh < 2; ++zh) {
            __syncthreads();                                        // the previous half fully consumed
            if constexpr (S4) {
                // ---- B: one thread per (sample, channel, halo plane, first halo row of a pair): two rows of 4 -> normalise, scale, split -> three shifts
                if (tid < 4 * NC * 30) {
                    const int sl = tid / (NC * 30), c = (tid / 30) % NC, r = tid % 30;
                    const int hz = r / 5, hy0 = r % 5;
                    const int sm = nn + 4 * zh + sl, z = hz - 1, ci = cchunk * NC + c;
                    _Float16 h[2][6], l[2][6];
#pragma unroll
                    for (int yy = 0; yy < 2; ++yy) {
                        const int y = hy0 + yy - 1;
#pragma unroll
                        for (int k = 0; k < 6; ++k) { h[yy][k] = (_Float16)0.f; l[yy][k] = (_Float16)0.f; }
                        if (ci < cin && sm < a.n && (unsigned)z < 4u && (unsigned)y < 4u) {
                            const float4 af = a.affine[(size_t)sm * cin + ci];
                            const float4 v = *reinterpret_cast<const float4*>(a.x + ((size_t)sm * cin + ci) * 64 + (z * 4 + y) * 4);
                            const float raw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (int k = 0; k < 4; ++k) ws_split(fmaf(raw[k] - af.x, af.y, af.z) * WS_ACT_SCALE, h[yy][k + 1], l[yy][k + 1]);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        h8 oh, ol;
#pragma unroll
                        for (int k = 0; k < 8; ++k) { oh[k] = h[k >> 2][t + (k & 3)]; ol[k] = l[k >> 2][t + (k & 3)]; }
                        Bh[tid * 3 + t] = oh;
                        Bl[tid * 3 + t] = ol;
                    }
                }
            } else {
                // ---- B: one thread per (channel, halo row): 10 values -> normalise, scale, split -> three shifted 8-voxel operands
                if (tid < NC * 60) {
                    const int c = tid / 60, r = tid % 60;
                    const int hz = r / 10, hy = r % 10;
                    const int z = z0 + 4 * zh + hz - 1, y = y0 + hy - 1, ci = cchunk * NC + c;
                    _Float16 h[10], l[10];
                    if (ci < cin && (unsigned)z < (unsigned)edge && (unsigned)y < (unsigned)edge) {
                        const float4 af = a.affine[(size_t)nn * cin + ci];
                        const float* row = a.x + ((size_t)nn * cin + ci) * vol + ((size_t)z * edge + y) * edge + x0;
                        const float4 v0 = *reinterpret_cast<const float4*>(row), v1 = *reinterpret_cast<const float4*>(row + 4);
                        const float vm = x0 > 0 ? row[-1] : 0.f, vp = x0 + 8 < edge ? row[8] : 0.f;
                        const float raw[10] = {vm, v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, vp};
#pragma unroll
                        for (int k = 0; k < 10; ++k) ws_split(fmaf(raw[k] - af.x, af.y, af.z) * WS_ACT_SCALE, h[k], l[k]);
                        if (x0 == 0) { h[0] = (_Float16)0.f; l[0] = (_Float16)0.f; }              // zero padding is of xn, not of x
                        if (x0 + 8 >= edge) { h[9] = (_Float16)0.f; l[9] = (_Float16)0.f; }
                    } else {
#pragma unroll
                        for (int k = 0; k < 10; ++k) { h[k] = (_Float16)0.f; l[k] = (_Float16)0.f; }
                    }
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        h8 oh, ol;
#pragma unroll
                        for (int k = 0; k < 8; ++k) { oh[k] = h[t + k]; ol[k] = l[t + k]; }
                        Bh[tid * 3 + t] = oh;
                        Bl[tid * 3 + t] = ol;
                    }
                }
            }
            // ---- A: (cout, row) items, four per thread: the row's 8 gradients, scaled, split
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int it = tid + k * 512, co = it >> 5, r = it & 31;
                h8 oh, ol;
                const bool live = cob + co < cout && (!S4 || nn + 4 * zh + (r >> 3) < a.n);
                if (live) {
                    const float* row = S4 ? a.dz + ((size_t)(nn + 4 * zh + (r >> 3)) * cout + cob + co) * 64 + (r & 7) * 8
                                          : a.dz + ((size_t)nn * cout + cob + co) * vol + ((size_t)(z0 + 4 * zh + (r >> 3)) * edge + (y0 + (r & 7))) * edge + x0;
                    const float4 v0 = *reinterpret_cast<const float4*>(row), v1 = *reinterpret_cast<const float4*>(row + 4);
                    const float raw[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                    for (int q = 0; q < 8; ++q) { _Float16 hh, ll; ws_split(raw[q] * sdz, hh, ll); oh[q] = hh; ol[q] = ll; }
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { oh[q] = (_Float16)0.f; ol[q] = (_Float16)0.f; }
                }
                Ah[co * WS_A_STRIDE + r] = oh;
                Al[co * WS_A_STRIDE + r] = ol;
            }
            __syncthreads();
            if (m_live) {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    // row r = 4 s + kg of the half.  Boxes: z = s >> 1, y = 4 (s & 1) + kg.  4^3: sample s >> 1, z = 2 (s & 1) + (kg >> 1), y pair kg & 1.
                    const int roff = S4 ? ((s >> 1) * NC * 30 + 2 * (s & 1) * 5) * 3 : ((s >> 1) * 10 + 4 * (s & 1)) * 3;
                    h8 ah[2], al[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) { ah[i] = Ah[abase[i] + 4 * s]; al[i] = Al[abase[i] + 4 * s]; }
#pragma unroll
                    for (int j = 0; j < JM; ++j) {
                        if (j < nj) {
                            const h8 bh = Bh[bbase[j] + roff], bl = Bl[bbase[j] + roff];
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                hi[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, hi[i][j], 0, 0, 0);
                                lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, lo[i][j], 0, 0, 0);
                                lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, lo[i][j], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
    }
    // D[row = cout 4 kg + r of the tile][col = li]
    if (m_live) {
#pragma unroll
        for (int j = 0; j < JM; ++j) {
            const int col = (wn + 4 * j) * 16 + li;
            if (j < nj && col < NC * 27) {
                const int ci = cchunk * NC + col / 27, tap = col % 27;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = cob + (2 * wm + i) * 16 + kg * 4 + r;
                        if (co < cout && ci < cin) a.parts[(((size_t)g * cout + co) * cin + ci) * 27 + tap] = fmaf(lo[i][j][r], 1.0f / WS_LO, hi[i][j][r]);
                    }
            }
        }
    }
}

// 64 outputs x 4 slices of the groups per workgroup: a thread sums its groups (g = slice, slice + 4, ...) in float64, the four partial sums are then
// added in slice order -- a fixed order whatever the launch (the first form, one thread per output walking all the groups, ran 27 workgroups for a
// 16 -> 16 layer and took as long as the MFMA kernel)
__global__ __launch_bounds__(256) void k_wgrad_split_reduce(const float* __restrict__ parts, int gb, size_t count, const float* __restrict__ scales,
                                                            float* __restrict__ dw) {
    __shared__ double red[4][64];
    const double back = (double)scales[1] / (double)WS_ACT_SCALE;   // 16 / s
    const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + o;
    double s = 0.0;
    if (i < count)
        for (int g = sl; g < gb; g += 4) s += (double)parts[(size_t)g * count + i];
    red[sl][o] = s;
    __syncthreads();
    if (sl == 0 && i < count) dw[i] = (float)((((red[0][o] + red[1][o]) + red[2][o]) + red[3][o]) * back);
}

static int wgrad_split_groups(int cin, int cout, int n, int edge) {
    const int nc = ws_nc(edge == 4);
    const long long boxes = edge == 4 ? (n + 7) / 8 : (long long)n * (edge / 8) * (edge / 8) * (edge / 8);
    const long long units = (long long)((cin + nc - 1) / nc) * ((cout + 63) / 64);
    long long gb = (768 + units - 1) / units;                       // about 768 workgroups in all: three rounds of one workgroup per CU
    if (gb > boxes) gb = boxes;
    return (int)(gb < 1 ? 1 : (gb > 256 ? 256 : gb));
}

extern "C" int rf_conv3d_k3_wgrad_split_supported(int cin, int cout, int n, int edge) {
    if (edge == 4) return cin >= 3 && 4 * cin >= 3 * rf_round_up(cin, 4) && cout >= 8 && n >= 8;
    return cin >= 6 && 4 * cin >= 3 * rf_round_up(cin, 8) && cout >= 8 && n > 0 && rf_is_pow2(edge) && edge >= 8 && edge <= 128;
}

extern "C" size_t rf_conv3d_k3_wgrad_split_ws_bytes(int cin, int cout, int n, int edge) {
    return (size_t)wgrad_split_groups(cin, cout, n, edge) * cout * cin * 27 * sizeof(float);
}

// x [n][cin][edge^3] (the layer input), gn_affine as the forward, dz [n][cout][edge^3], scales = (s, 1 / s) with |dz| * s < 65504 (rf_dgrad_scale_affine)
// -> dw OIDHW [cout][cin][27]
extern "C" int rf_conv3d_k3_wgrad_split(const float* x, int cin, int n, int edge, const float* gn_affine, const float* dz, int cout, const float* scales,
                                        float* dw, void* ws, size_t ws_bytes, void* stream) {
    RF_REQUIRE(x && gn_affine && dz && scales && dw && ws, RF_E_INVALID, "rf_conv3d_k3_wgrad_split: null pointer");
    RF_REQUIRE(rf_conv3d_k3_wgrad_split_supported(cin, cout, n, edge), RF_E_UNSUPPORTED,
               "rf_conv3d_k3_wgrad_split: takes cin >= 6 (at least 3/4 of the next multiple of 8; 4^3: of 4), cout >= 8, edge a power of two >= 8 or 8+ whole 4^3 samples (got cin=%d cout=%d n=%d edge=%d)",
               cin, cout, n, edge);
    RF_REQUIRE(ws_bytes >= rf_conv3d_k3_wgrad_split_ws_bytes(cin, cout, n, edge), RF_E_WORKSPACE, "rf_conv3d_k3_wgrad_split: workspace too small");
    static RfLdsOptIn opt_box, opt_s4;
    WgradSplitArgs a;
    a.x = x; a.affine = reinterpret_cast<const float4*>(gn_affine); a.dz = dz; a.scales = scales; a.parts = (float*)ws;
    a.cin = cin; a.cout = cout; a.n = n; a.edge = edge; a.gb = wgrad_split_groups(cin, cout, n, edge);
    hipStream_t s = (hipStream_t)stream;
    const int nc = ws_nc(edge == 4);
    const dim3 grid((cin + nc - 1) / nc, (cout + 63) / 64, a.gb);
    if (edge == 4) {
        if (int rc = opt_s4.ensure(reinterpret_cast<const void*>(k_conv3_wgrad_split<true>), WS_LDS_BYTES, "rf_conv3d_k3_wgrad_split")) return rc;
        hipLaunchKernelGGL(k_conv3_wgrad_split<true>, grid, dim3(512), WS_LDS_BYTES, s, a);
    } else {
        if (int rc = opt_box.ensure(reinterpret_cast<const void*>(k_conv3_wgrad_split<false>), WS_LDS_BYTES, "rf_conv3d_k3_wgrad_split")) return rc;
        hipLaunchKernelGGL(k_conv3_wgrad_split<false>, grid, dim3(512), WS_LDS_BYTES, s, a);
    }
    RF_CHECK_LAUNCH("rf_conv3d_k3_wgrad_split");
    const size_t count = (size_t)cout * cin * 27;
    hipLaunchKernelGGL(k_wgrad_split_reduce, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, s, (const float*)ws, a.gb, count, scales, dw);
    RF_CHECK_LAUNCH("rf_conv3d_k3_wgrad_split(reduce)");
    return RF_OK;
}
