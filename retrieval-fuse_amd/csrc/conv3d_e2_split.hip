// rf_conv3d_e2_split_k3_gn_relu: SingleConv 'gcr' (reference model/unet.py:19-76: GroupNorm -> 3x3x3 conv, pad 1, no bias -> ReLU) on whole 2^3 volumes
// (and 1^3 volumes: the same GEMM with K = cin, N = cout, the centre tap)
// (the deepest level of the retrieval backbone: 64 -> 64 and 64 -> 128 on 8192 patches per step of C2) as ONE DENSE GEMM on the F16 matrix cores.
//
// In a 2^3 volume every input voxel u is a neighbour of every output voxel v (|u - v| <= 1 in each dimension), so
//     out[n][co][v] = sum_{ci, u} xn[n][ci][u] * W[co][ci][tap(u - v)]            tap(d) = ((dz + 1) * 3 + dy + 1) * 3 + dx + 1
// is  Y[n][(co, v)] = XN[n][(ci, u)] . B[(ci, u)][(co, v)]  with M = samples, K = 8 cin, N = 8 cout -- and both X [n][cin][2][2][2] and Y
// [n][cout][2][2][2] ARE those row-major matrices: no gather, no halo, no zero-padding tap (64 of the 27 x 8 = 216 (tap, voxel) pairs a box kernel would
// issue are real).  Arithmetic of conv3d_split.hip: operands as f16 pairs (x = h + l / 2^11; activations scaled 2^-4, weights 2^4), exact
// f16 x f16 products, hi / lo fp32 accumulators, 3 MFMAs per product tile.
//
// A k-step (k = 32) is 4 channels x 8 voxels: lane (row = sample, kg) loads channel 4s + kg of its sample (8 contiguous floats), applies the
// GroupNorm affine of (sample, channel), splits -- A operands never touch LDS.  B is pre-packed in fragment order ([n-chunk of 256 columns][k-step]
// [16 n-blocks][h | l][lane]: 32 KB per k-step, rf_conv3_e2_split_pack_weight) and staged through LDS (two buffers) for the workgroup's four waves.
// Workgroup = 64 samples x 256 columns, 4 waves as 2 (32 samples) x 2 (128 columns); a wave holds 2 x 8 tiles, hi and lo (128 accumulator
// registers): 16 LDS operand reads + 2 conversions per 48 MFMAs.  Epilogue: ReLU, rows of 16 consecutive columns (64 bytes) straight to memory,
// GroupNorm statistics of the output (sum, sum of squares per (sample, cout): the 8 voxels of a cout are 8 neighbouring lanes) in float64.
// The fp32 position-major kernels this replaces (k_conv3_mfma<2,2,2,16>, k_conv3_small<2,2>) took 125 / 94 us per launch on 8192 samples.
#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int E2_NC = 256;                                   // columns per workgroup (32 couts)
constexpr int E2_STEP_H8 = (E2_NC / 16) * 2 * 64;            // h8 elements of one k-step of one n-chunk: 2048 = 32 KB
constexpr int E2_LDS_BYTES = 2 * E2_STEP_H8 * 16;            // 65,536
constexpr float E2_ACT_SCALE = 1.0f / 16, E2_W_SCALE = 16.0f, E2_LO = 2048.0f;
}

// edge = 2: K = 8 cin, N = 8 cout (k-step = 4 channels x 8 voxels).  edge = 1: only the centre tap touches data: K = cin, N = cout (k-step = 32 channels).
__host__ __device__ static inline size_t e2_chunks(int cout, int edge) { return ((size_t)cout * (edge == 2 ? 8 : 1) + E2_NC - 1) / E2_NC; }
__host__ __device__ static inline size_t e2_ksteps(int cin, int edge) { return edge == 2 ? (size_t)(cin + 3) / 4 : (size_t)(cin + 31) / 32; }

extern "C" size_t rf_conv3_e2_split_packed_bytes(int cout, int cin, int edge) {
    return e2_chunks(cout, edge) * e2_ksteps(cin, edge) * E2_STEP_H8 * 16;
}

// wp[chunk][k-step s][n-block t][piece][lane]: lane (li, kg) holds K = 32 s + 8 kg + j of column chunk * 256 + t * 16 + li.
// edge 2: K -> (ci = 4 s + kg, u = j), column -> (co, v);  edge 1: K -> ci = 32 s + 8 kg + j, column -> co
__global__ void k_conv3_e2_split_pack(const float* __restrict__ w, int cout, int cin, int edge, h8* __restrict__ wp, size_t total) {
    const int ksteps = (int)e2_ksteps(cin, edge);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), piece = (int)((i >> 6) & 1), t = (int)((i >> 7) & 15);
        const size_t st = i >> 11;
        const int s = (int)(st % ksteps), chunk = (int)(st / ksteps);
        const int col = chunk * E2_NC + t * 16 + (lane & 15);
        h8 out;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            double val = 0.0;
            if (edge == 2) {
                const int co = col >> 3, v = col & 7, ci = 4 * s + (lane >> 4);
                if (co < cout && ci < cin) {
                    const int dz = (u >> 2) - (v >> 2), dy = ((u >> 1) & 1) - ((v >> 1) & 1), dx = (u & 1) - (v & 1);
                    val = (double)w[((size_t)co * cin + ci) * 27 + ((dz + 1) * 3 + dy + 1) * 3 + dx + 1];
                }
            } else {
                const int co = col, ci = 32 * s + 8 * (lane >> 4) + u;
                if (co < cout && ci < cin) val = (double)w[((size_t)co * cin + ci) * 27 + 13];
            }
            val *= (double)E2_W_SCALE;
            val = val > 65504.0 ? 65504.0 : (val < -65504.0 ? -65504.0 : val);
            const _Float16 h = (_Float16)(float)val;
            out[u] = piece == 0 ? h : (_Float16)(float)((val - (double)(float)h) * (double)E2_LO);
        }
        wp[i] = out;
    }
}

extern "C" int rf_conv3_e2_split_pack_weight(const float* w_oidhw, int cout, int cin, int edge, void* w_packed, void* stream) {
    RF_REQUIRE(w_oidhw && w_packed && cout > 0 && cin > 0 && (edge == 1 || edge == 2), RF_E_INVALID, "rf_conv3_e2_split_pack_weight: bad arguments");
    const size_t total = rf_conv3_e2_split_packed_bytes(cout, cin, edge) / 16, want = (total + 255) / 256;
    hipLaunchKernelGGL(k_conv3_e2_split_pack, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream, w_oidhw, cout, cin, edge,
                       reinterpret_cast<h8*>(w_packed), total);
    RF_CHECK_LAUNCH("rf_conv3_e2_split_pack_weight");
    return RF_OK;
}

struct E2Args {
    const float* src;          // [n][cin][8]
    const float4* affine;      // [n][cin] (mean, scale, shift, -)
    const h8* wp;
    float* out;                // [n][cout][8]
    double2* stats;            // [n][cout] (sum, sum of squares) or null
    int cin, cout, n;
};

// MT: samples per workgroup.  64: 4 waves as 2 (32 samples) x 2 (128 columns).  32: 4 waves x 64 columns, all 32 samples -- twice the workgroups for launches
// that would otherwise put ONE workgroup (one wave per SIMD) on a CU: the k-step chain (B block global -> registers -> LDS -> barrier -> MFMAs) is a
// latency chain, and a lone wave per SIMD has nothing to run while it waits (64 -> 64 on 8192 samples: 256 workgroups, 61 us for 6 us of MFMAs).
template <int V, int MT>      // V: voxels per volume: 8 (edge 2) or 1 (edge 1)
__global__ __launch_bounds__(256, 2) void k_conv3_e2_split(E2Args a) {
    constexpr int WN = MT == 64 ? 2 : 4, NJ = 16 / WN;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    h8* bufs = reinterpret_cast<h8*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = MT == 64 ? wave >> 1 : 0, wn = MT == 64 ? wave & 1 : wave;
    const int li = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * MT, chunk = blockIdx.y;
    const int cin = a.cin, cout = a.cout, ksteps = V == 8 ? cin >> 2 : (cin + 31) >> 5;
    const h8* __restrict__ wsrc = a.wp + (size_t)chunk * ksteps * E2_STEP_H8 + tid;

    // this lane's two A rows (samples); rows past n read sample n - 1 and are never stored
    const float* xrow[2];
    const float4* arow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int sm = n0 + (2 * wm + i) * 16 + li;
        sm = sm < a.n ? sm : a.n - 1;
        xrow[i] = a.src + (V == 8 ? ((size_t)sm * cin + kg) * 8 : (size_t)sm * cin + kg * 8);
        arow[i] = a.affine + (V == 8 ? (size_t)sm * cin + kg : (size_t)sm * cin + kg * 8);
    }
    f32x4 hi[2][NJ], lo[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) { hi[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    float4 xa[2][2], af[2];                                       // edge 2: the raw row and its affine, converted after the MFMAs they were loaded under
    h8 nah[2], nal[2];                                             // edge 1: converted at once (8 affines per row would not fit beside the accumulators)
    h8 wreg[8];
    auto split1 = [](float y, _Float16& hh, _Float16& ll) {
        const float v = __builtin_amdgcn_fmed3f(y * E2_ACT_SCALE, -65504.f, 65504.f);
        hh = (_Float16)v;
        ll = (_Float16)fmaf(-E2_LO, (float)hh, v * E2_LO);
    };
    auto load_step = [&](int s) {                                  // A rows and the B block of k-step s
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (V == 8) {
                xa[i][0] = *reinterpret_cast<const float4*>(xrow[i] + (size_t)s * 32);
                xa[i][1] = *reinterpret_cast<const float4*>(xrow[i] + (size_t)s * 32 + 4);
                af[i] = arow[i][(size_t)s * 4];
            } else {                                                // 8 channels of the one voxel; channels past cin: zeros (their weights are zero too)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool ok = 32 * s + 8 * kg + j < cin;
                    const float x = ok ? xrow[i][(size_t)s * 32 + j] : 0.f;
                    const float4 t = ok ? arow[i][(size_t)s * 32 + j] : make_float4(0.f, 0.f, 0.f, 0.f);
                    _Float16 hh, ll;
                    split1(fmaf(x - t.x, t.y, t.z), hh, ll);
                    nah[i][j] = hh; nal[i][j] = ll;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) wreg[q] = wsrc[(size_t)s * E2_STEP_H8 + q * 256];
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 8; ++q) bufs[buf * E2_STEP_H8 + q * 256 + tid] = wreg[q];
    };
    load_step(0);
    store_b(0);
    h8 ah[2], al[2];
    auto convert = [&] {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (V == 8) {
                const float raw[8] = {xa[i][0].x, xa[i][0].y, xa[i][0].z, xa[i][0].w, xa[i][1].x, xa[i][1].y, xa[i][1].z, xa[i][1].w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    _Float16 hh, ll;
                    split1(fmaf(raw[j] - af[i].x, af[i].y, af[i].z), hh, ll);
                    ah[i][j] = hh; al[i][j] = ll;
                }
            } else {
                ah[i] = nah[i]; al[i] = nal[i];
            }
        }
    };
    convert();
    __syncthreads();

    for (int s = 0; s < ksteps; ++s) {
        const bool more = s + 1 < ksteps;
        if (more) load_step(s + 1);                                 // in flight under this step's MFMAs
        const h8* bb = bufs + (s & 1) * E2_STEP_H8 + (wn * NJ) * 128 + lane;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const h8 bh = bb[j * 128], bl = bb[j * 128 + 64];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                hi[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, hi[i][j], 0, 0, 0);
                lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, lo[i][j], 0, 0, 0);
                lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, lo[i][j], 0, 0, 0);
            }
        }
        if (more) {
            store_b((s + 1) & 1);                                   // the other buffer: its last readers passed the barrier of step s - 1
            convert();
        }
        __syncthreads();
    }

    // D[row = sample 4 kg + r of the m-block][col = li]: column = chunk * 256 + (wn * 8 + j) * 16 + li -> (co, v) (edge 1: co)
    const int ncols = cout * V;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = chunk * E2_NC + (wn * NJ + j) * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int sm = n0 + (2 * wm + i) * 16 + kg * 4 + r;
                const float y = fmaxf(fmaf(lo[i][j][r], 1.0f / E2_LO, hi[i][j][r]), 0.f);
                const bool live = sm < a.n && col < ncols;
                if (live) a.out[(size_t)sm * ncols + col] = y;
                if (a.stats) {                                      // the 8 voxels of a cout: lanes li & 7 = 0 .. 7 of the same row (wave-uniform branch)
                    double sv = (double)y, sq = (double)y * (double)y;
                    if constexpr (V == 8) {
#pragma unroll
                        for (int d = 1; d <= 4; d <<= 1) { sv += __shfl_xor(sv, d, 64); sq += __shfl_xor(sq, d, 64); }
                    }
                    if (live && (V == 1 || (li & 7) == 0)) a.stats[(size_t)sm * cout + (V == 8 ? col >> 3 : col)] = make_double2(sv, sq);
                }
            }
        }
}

extern "C" int rf_conv3d_e2_split_supported(int cin, int n, int edge, int cout) {
    // from 16 samples on: below that a workgroup's 64 rows are nearly all padding and the fp32 position-major / direct kernels take the call
    if (edge == 1) return cin >= 8 && cout >= 2 && n >= 16;
    return edge == 2 && cin >= 8 && cin % 4 == 0 && cout >= 2 && n >= 16;
}

// src [n][cin][edge^3], gn_affine [n][cin][4] (mean, scale, shift, -), w_packed from rf_conv3_e2_split_pack_weight (same edge) -> out [n][cout][edge^3]
// (ReLU'd) and, optionally, stats [n][cout][1 tile][2] float64 (sum, sum of squares) -- the layout of the other conv entry points' statistics with one tile
extern "C" int rf_conv3d_e2_split_k3_gn_relu(const float* src, int cin, int n, int edge, const float* gn_affine, const void* w_packed, int cout,
                                              float* out, double* stats, void* stream) {
    RF_REQUIRE(rf_conv3d_e2_split_supported(cin, n, edge, cout), RF_E_UNSUPPORTED,
               "rf_conv3d_e2_split_k3_gn_relu: takes whole 2^3 volumes (cin a multiple of 4) or 1^3 volumes, cin >= 8, at least 16 samples (got cin=%d n=%d edge=%d cout=%d)", cin, n, edge, cout);
    RF_REQUIRE(src && gn_affine && w_packed && out, RF_E_INVALID, "rf_conv3d_e2_split_k3_gn_relu: null pointer");
    E2Args a;
    a.src = src; a.affine = reinterpret_cast<const float4*>(gn_affine); a.wp = reinterpret_cast<const h8*>(w_packed); a.out = out;
    a.stats = reinterpret_cast<double2*>(stats); a.cin = cin; a.cout = cout; a.n = n;
    // 64-sample workgroups when they still give every CU two workgroups, else 32-sample ones
    const unsigned chunks = (unsigned)e2_chunks(cout, edge);
    const bool wide = (long long)((n + 63) / 64) * chunks >= rf_persistent_wgs() / RF_PERSIST_ROUNDS;
    const dim3 grid((unsigned)((n + (wide ? 63 : 31)) / (wide ? 64 : 32)), chunks);
    static RfLdsOptIn opt[4];
    auto launch = [&](auto kern, int slot) -> int {
        if (int rc = opt[slot].ensure(reinterpret_cast<const void*>(kern), E2_LDS_BYTES, "rf_conv3d_e2_split_k3_gn_relu")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(256), E2_LDS_BYTES, (hipStream_t)stream, a);
        return RF_OK;
    };
    int rc;
    if (edge == 2) rc = wide ? launch(k_conv3_e2_split<8, 64>, 0) : launch(k_conv3_e2_split<8, 32>, 1);
    else rc = wide ? launch(k_conv3_e2_split<1, 64>, 2) : launch(k_conv3_e2_split<1, 32>, 3);
    if (rc != RF_OK) return rc;
    RF_CHECK_LAUNCH("rf_conv3d_e2_split_k3_gn_relu");
    return RF_OK;
}
