// Volume-domain patch attention for gfx950 (attention patch extent e = 2, hidden width 128, feature width 32: every
// shipped config).  Replaces, for PatchedAttentionBlock.forward (reference model/attention.py:141-157), the chain
//   Unfold3D -> regroup of the retrieved features -> 2 x AttentionFeatureEncoder (4 Linear layers each, :29-46)
//   -> AttentionBlock.forward (:84-113) -> Fold3D
// by three kernels that never materialise the unfolded rows:
//
//   k_attn_mlp      theta / phi encoder: the 4 Linear layers + LeakyReLU(0.01) fused, fp32 MFMA, activations never leave
//                   registers.  The rows are read straight out of the NCDHW volume / the retrieval backbone's patch-major
//                   output (two 8-byte loads per 4 features).
//   k_attn_weights  per row: normalise, K scores, switch, softmax | Gumbel-hard weights          (same arithmetic as k_attn_fuse)
//   k_attn_blend    out[b][c][z][y][x] = x*(1-switch) + (sum_k w_k * retrieved_k)*switch, in the folded layout
//
// MLP on the matrix cores without a transpose between layers: a layer is computed TRANSPOSED, D[feature][row] =
// W[feature][k] . H^T[k][row], with v_mfma_f32_16x16x4_f32 (A = weights, B = activations).  A lane of D holds, for row
// j = lane&15, the 4 consecutive features 4g..4g+3 (g = lane>>4) of a 16-feature block -- and the B operand of the next
// layer wants, for k-step r, lane (g, j) to supply H[row j][some k].  Choosing the contraction order k <-> feature
// 16*kb + 4*g + r makes that exactly register r of the previous D: the weight image is packed in the matching order
// ([kb][ib][lane][r] = W[16*ib + (lane&15)][16*kb + 4*(lane>>4) + r]) and the activations are consumed where they are.
#include "common.h"
#include "attn_row.h"

// The scalar arithmetic here restates torch expressions op by op (every product and sum rounded).  This file is built
// with -ffp-contract=off (csrc/build.py): hipcc's default -ffp-contract=fast fuses a*b+c in the backend, where neither
// __fmul_rn/__fadd_rn nor `#pragma clang fp contract(off)` reach.  Explicit fmaf() stays an FMA.

typedef __attribute__((address_space(1))) const void* rf_gptr;
typedef __attribute__((address_space(3))) void* rf_lptr;

#define AM_HID 128
#define AM_OUT 32
#define AM_WAVES 16
#define AM_BUF_FLOATS (AM_HID * AM_HID)          // one layer's weights: 64 KiB

// ------------------------------------------------------------------------------------------------ weight image
static __host__ __device__ inline size_t am_layer_off(int n_in, int layer) {
    size_t o = 0;
    if (layer > 0) o += (size_t)n_in * AM_HID;
    if (layer > 1) o += (size_t)AM_HID * AM_HID;
    if (layer > 2) o += (size_t)AM_HID * AM_HID;
    if (layer > 3) o += (size_t)AM_HID * AM_OUT;                 // start of the biases
    return o;
}
static __host__ __device__ inline size_t am_bias_off(int n_in, int layer) { return am_layer_off(n_in, 4) + (size_t)layer * AM_HID; }

extern "C" size_t rf_attn_mlp_packed_floats(int n_in) { return am_bias_off(n_in, 3) + AM_OUT; }

struct AmPackArgs {
    const float* w[4];
    const float* b[4];
    float* img;
    int n_in;
};

__global__ void k_attn_mlp_pack(AmPackArgs a) {
    const size_t total = am_bias_off(a.n_in, 3) + AM_OUT;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float v;
        if (i >= am_layer_off(a.n_in, 4)) {
            const size_t bi = i - am_layer_off(a.n_in, 4);
            const int layer = (int)(bi / AM_HID) > 3 ? 3 : (int)(bi / AM_HID);
            v = a.b[layer][bi - (size_t)layer * AM_HID];
        } else {
            int layer = 0;
            while (layer < 3 && i >= am_layer_off(a.n_in, layer + 1)) ++layer;
            const size_t li = i - am_layer_off(a.n_in, layer);
            const int nin = layer == 0 ? a.n_in : AM_HID;
            const int ibn = layer == 3 ? AM_OUT / 16 : AM_HID / 16;
            const int r = (int)(li & 3), lane = (int)((li >> 2) & 63);
            const int ib = (int)((li >> 8) % ibn), kb = (int)((li >> 8) / ibn);
            v = a.w[layer][(size_t)(ib * 16 + (lane & 15)) * nin + kb * 16 + 4 * (lane >> 4) + r];
        }
        a.img[i] = v;
    }
}

extern "C" int rf_attn_mlp_pack(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                                const float* w4, const float* b4, int n_in, float* packed, void* stream) {
    RF_REQUIRE(w1 && b1 && w2 && b2 && w3 && b3 && w4 && b4 && packed, RF_E_INVALID, "rf_attn_mlp_pack: null pointer");
    RF_REQUIRE(n_in >= 16 && n_in <= AM_HID && n_in % 16 == 0, RF_E_UNSUPPORTED, "rf_attn_mlp_pack: n_in %d must be a multiple of 16 in 16..128", n_in);
    AmPackArgs a;
    a.w[0] = w1; a.w[1] = w2; a.w[2] = w3; a.w[3] = w4;
    a.b[0] = b1; a.b[1] = b2; a.b[2] = b3; a.b[3] = b4;
    a.img = packed; a.n_in = n_in;
    hipLaunchKernelGGL(k_attn_mlp_pack, dim3(256), dim3(256), 0, (hipStream_t)stream, a);
    RF_CHECK_LAUNCH("rf_attn_mlp_pack");
    return RF_OK;
}

// ------------------------------------------------------------------------------------------------ fused MLP
struct AmArgs {
    const float* src;
    const float* img;
    const float* img2;     // split-operand image (rf_attn_mlp_split_pack) or NULL: fp32 MFMA form
    float* out;            // [out rows][32]
    int mode;              // 0: src = rows [nrows][n_in]; 1: src = volumes / patch-major features, e = 2
    int nrows, n_in, ntiles;
    int kv, c, s, t;       // mode 1: vol = bb*kv + k, channels, volume edge, source patch edge (t == s: whole NCDHW volumes)
};

// one layer, transposed: acc[ib] (+)= W[ib-block][kb-block] . hin[kb]; weights from LDS (one 16-byte read per 4 MFMAs)
template <int IB>
__device__ __forceinline__ void am_layer(const float* wbuf, int kbn, const f32x4 (&hin)[8], f32x4 (&acc)[8], int lane) {
    constexpr int GRP = 2;                                         // independent accumulators between dependent MFMAs
    constexpr int NG = IB / GRP;
    const float4* wv = reinterpret_cast<const float4*>(wbuf) + lane;
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) acc[ib] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 aw[2][GRP];
#pragma unroll
    for (int i = 0; i < GRP; ++i) aw[0][i] = wv[i * 64];
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        if (kb < kbn) {                                            // uniform
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                const int st = kb * NG + gi, cur = st & 1, nxt = cur ^ 1;
                if (st + 1 < 8 * NG) {                             // next group's weights fly under this group's MFMAs
#pragma unroll
                    for (int i = 0; i < GRP; ++i) aw[nxt][i] = wv[((st + 1) * GRP + i) * 64];
                }
#pragma unroll
                for (int i = 0; i < GRP; ++i) acc[gi * GRP + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[cur][i].x, hin[kb][0], acc[gi * GRP + i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < GRP; ++i) acc[gi * GRP + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[cur][i].y, hin[kb][1], acc[gi * GRP + i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < GRP; ++i) acc[gi * GRP + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[cur][i].z, hin[kb][2], acc[gi * GRP + i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < GRP; ++i) acc[gi * GRP + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[cur][i].w, hin[kb][3], acc[gi * GRP + i], 0, 0, 0);
            }
        }
    }
}

__global__ __launch_bounds__(AM_WAVES * 64) void k_attn_mlp(AmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];     // two weight buffers of AM_BUF_FLOATS
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int kb0 = a.n_in >> 4;

    // weights of `layer` -> LDS buffer layer & 1, by DMA (1-KiB pieces round-robin over the waves)
    auto dma_layer = [&](int layer) {
        const int npiece = layer == 0 ? kb0 * 8 : (layer == 3 ? (AM_HID / 16) * (AM_OUT / 16) : 64);
        const float* src = a.img + am_layer_off(a.n_in, layer);
        float* dst = smem + (layer & 1) * AM_BUF_FLOATS;
        for (int q = wave; q < npiece; q += AM_WAVES)
            __builtin_amdgcn_global_load_lds((rf_gptr)(src + q * 256 + lane * 4), (rf_lptr)(dst + q * 256), 16, 0, 0);
    };
    const float4* bias4 = reinterpret_cast<const float4*>(a.img + am_layer_off(a.n_in, 4));     // [layer][32 float4s]

    dma_layer(0);
    const int nwt = (a.ntiles + AM_WAVES - 1) / AM_WAVES;             // workgroup tiles of 16 waves x 16 rows
    for (int wt = blockIdx.x; wt < nwt; wt += gridDim.x) {
        const int rt = wt * AM_WAVES + wave;
        const bool live = rt < a.ntiles;
        // ---- this lane's row and where its input lives
        int row = rt * 16 + j;
        if (row >= a.nrows) row = a.nrows - 1;                        // clamp: computed, never stored
        if (row < 0) row = 0;
        size_t orow = (size_t)row;
        f32x4 hin[8];
        if (a.mode == 0) {
            const float* p = a.src + (size_t)row * a.n_in + 4 * g;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                if (kb < kb0) {
                    const float4 v = *reinterpret_cast<const float4*>(p + kb * 16);
                    hin[kb] = (f32x4){v.x, v.y, v.z, v.w};
                } else {
                    hin[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        } else {
            const int r = a.s >> 1, t = a.t, q = a.s / t;
            const int r3 = r * r * r;
            const int vol = row / r3, prow = row - vol * r3;
            const int p2 = prow % r, p1 = (prow / r) % r, p0 = prow / (r * r);
            const int bb = vol / a.kv, k = vol - bb * a.kv;
            orow = ((size_t)bb * r3 + prow) * a.kv + k;
            const int d0 = 2 * p0 + (g & 1), d1 = 2 * p1, d2 = 2 * p2;
            const size_t t3 = (size_t)t * t * t;
            const size_t patch = (((size_t)vol * q + d0 / t) * q + d1 / t) * q + d2 / t;
            const float* p = a.src + (patch * a.c + (g >> 1)) * t3 + ((size_t)(d0 % t) * t + (d1 % t)) * t + (d2 % t);
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                if (kb < kb0) {                                        // features 16kb+4g+r: channel 2kb+(g>>1), e0 = g&1, (e1,e2) = r
                    const float2 lo = *reinterpret_cast<const float2*>(p + (size_t)(2 * kb) * t3);
                    const float2 hi = *reinterpret_cast<const float2*>(p + (size_t)(2 * kb) * t3 + t);
                    hin[kb] = (f32x4){lo.x, lo.y, hi.x, hi.y};
                } else {
                    hin[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        }

        f32x4 acc[8];
#pragma unroll
        for (int layer = 0; layer < 4; ++layer) {
            __syncthreads();                                          // weights of `layer` landed; everyone left layer-1's buffer
            // next weights: the following layer, or layer 0 of this workgroup's next tile
            if (layer < 3) dma_layer(layer + 1);
            else if (wt + (int)gridDim.x < nwt) dma_layer(0);
            const float* wbuf = smem + (layer & 1) * AM_BUF_FLOATS;
            if (layer < 3) {
                am_layer<8>(wbuf, layer == 0 ? kb0 : 8, hin, acc, lane);
#pragma unroll
                for (int ib = 0; ib < 8; ++ib) {                       // bias + LeakyReLU(0.01): the next layer's B operands
                    const float4 bz = bias4[layer * 32 + ib * 4 + g];
                    f32x4 v = acc[ib];
                    v[0] += bz.x; v[1] += bz.y; v[2] += bz.z; v[3] += bz.w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * 0.01f;
                    hin[ib] = v;
                }
            } else {
                am_layer<2>(wbuf, 8, hin, acc, lane);
                if (live && rt * 16 + j < a.nrows) {
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) {
                        const float4 bz = bias4[3 * 32 + ib * 4 + g];
                        const float4 o = make_float4(acc[ib][0] + bz.x, acc[ib][1] + bz.y, acc[ib][2] + bz.z, acc[ib][3] + bz.w);
                        *reinterpret_cast<float4*>(a.out + orow * AM_OUT + ib * 16 + 4 * g) = o;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ fused MLP, split-operand form
// The same encoder on the F16 matrix cores with every fp32 operand carried as two f16 pieces (x = h + l / 2^11, exact products, hi / lo
// fp32 accumulators: csrc/conv3d_up_split.hip has the numerics).  A k-step of v_mfma_f32_16x16x32_f16 is 32 input features; lane group
// g supplies 8 of them and holds, after a layer, 4 features of every 16-feature output block -- so k-step t is made of the output
// blocks 2t and 2t+1 (contraction order k <-> {16*(2t) + 4g + r, 16*(2t+1) + 4g + r}), the weight image is packed in that order, and
// the activations again never leave registers: bias, LeakyReLU and the split run on the D registers in place.
typedef _Float16 am_h8 __attribute__((ext_vector_type(8)));
#define AMS_ACT 0.0625f
#define AMS_W 16.0f
#define AMS_LO 2048.0f

// split image, in 16-byte fragments rows of 64 lanes: layer L at ams_layer_off(L): [t][ib][h|l][lane]
static __host__ __device__ inline int ams_steps(int n_in, int layer) { return layer == 0 ? (n_in / 16 + 1) / 2 : 4; }
static __host__ __device__ inline int ams_ibn(int layer) { return layer == 3 ? AM_OUT / 16 : AM_HID / 16; }
static __host__ __device__ inline size_t ams_layer_off(int n_in, int layer) {       // in floats (256 floats = one fragment row)
    size_t o = 0;
    for (int l = 0; l < layer; ++l) o += (size_t)ams_steps(n_in, l) * ams_ibn(l) * 2 * 256;
    return o;
}
extern "C" size_t rf_attn_mlp_split_packed_floats(int n_in) { return ams_layer_off(n_in, 4); }

struct AmsPackArgs {
    const float* w[4];
    float* img;
    int n_in;
};

__global__ void k_attn_mlp_split_pack(AmsPackArgs a) {
    const size_t total = ams_layer_off(a.n_in, 4) / 4;              // 16-byte entries
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int layer = 0;
        while (layer < 3 && i * 4 >= ams_layer_off(a.n_in, layer + 1)) ++layer;
        const size_t li = i - ams_layer_off(a.n_in, layer) / 4;
        const int nin = layer == 0 ? a.n_in : AM_HID, ibn = ams_ibn(layer);
        const int lane = (int)(li & 63), piece = (int)((li >> 6) & 1);
        const int ib = (int)((li >> 7) % ibn), t = (int)((li >> 7) / ibn);
        const int feat = ib * 16 + (lane & 15), g = lane >> 4;
        am_h8 out;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * (2 * t + (j >> 2)) + 4 * g + (j & 3);
            double v = k < nin ? (double)a.w[layer][(size_t)feat * nin + k] * (double)AMS_W : 0.0;
            v = v > 65504.0 ? 65504.0 : (v < -65504.0 ? -65504.0 : v);
            const _Float16 h = (_Float16)(float)v;
            out[j] = piece == 0 ? h : (_Float16)(float)((v - (double)(float)h) * (double)AMS_LO);
        }
        reinterpret_cast<am_h8*>(a.img)[i] = out;
    }
}

extern "C" int rf_attn_mlp_split_pack(const float* w1, const float* w2, const float* w3, const float* w4, int n_in, float* packed, void* stream) {
    RF_REQUIRE(w1 && w2 && w3 && w4 && packed, RF_E_INVALID, "rf_attn_mlp_split_pack: null pointer");
    RF_REQUIRE(n_in >= 16 && n_in <= AM_HID && n_in % 16 == 0, RF_E_UNSUPPORTED, "rf_attn_mlp_split_pack: n_in %d must be a multiple of 16 in 16..128", n_in);
    AmsPackArgs a;
    a.w[0] = w1; a.w[1] = w2; a.w[2] = w3; a.w[3] = w4; a.img = packed; a.n_in = n_in;
    hipLaunchKernelGGL(k_attn_mlp_split_pack, dim3(256), dim3(256), 0, (hipStream_t)stream, a);
    RF_CHECK_LAUNCH("rf_attn_mlp_split_pack");
    return RF_OK;
}

// 4 activation values (one D register quartet) -> 4 halves of the h and of the l piece, at positions o..o+3
__device__ __forceinline__ void ams_split4(const f32x4& v, am_h8& h, am_h8& l, int o) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = __builtin_amdgcn_fmed3f(v[e] * AMS_ACT, -65504.f, 65504.f);
        const _Float16 hh = (_Float16)x;
        h[o + e] = hh;
        l[o + e] = (_Float16)fmaf(-AMS_LO, (float)hh, x * AMS_LO);      // (x - h) * 2^11: exact either way, one v_fma_mix instead of cvt + sub + mul
    }
}

// one layer, transposed: hi/lo[ib] = W[ib-block][k-step t] . (bh, bl)[t]; weight fragments from LDS, one (t, ib) pair ahead
template <int IB>
__device__ __forceinline__ void ams_layer(const float* wbuf, int tn, const am_h8 (&bh)[4], const am_h8 (&bl)[4], f32x4 (&hi)[8], f32x4 (&lo)[8], int lane) {
    const am_h8* wv = reinterpret_cast<const am_h8*>(wbuf) + lane;
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) { hi[ib] = (f32x4){0.f, 0.f, 0.f, 0.f}; lo[ib] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    am_h8 wh[2], wl[2];
    wh[0] = wv[0]; wl[0] = wv[64];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t < tn) {                                                  // uniform
#pragma unroll
            for (int ib = 0; ib < IB; ++ib) {
                const int st = t * IB + ib, cur = st & 1, nxt = cur ^ 1;
                if (st + 1 < 4 * IB) { wh[nxt] = wv[(st + 1) * 128]; wl[nxt] = wv[(st + 1) * 128 + 64]; }
                hi[ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[cur], bh[t], hi[ib], 0, 0, 0);
                lo[ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[cur], bl[t], lo[ib], 0, 0, 0);
                lo[ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[cur], bh[t], lo[ib], 0, 0, 0);
            }
        }
    }
}

// 12 waves (three per SIMD: 168 registers each) instead of the fp32 kernel's 16: at 128 registers the layer loop spilled 22 of them (270 -> 250 us on 524 288 rows)
#define AMS_WAVES 12
__global__ __launch_bounds__(AMS_WAVES * 64) void k_attn_mlp_split(AmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];     // two weight buffers of AM_BUF_FLOATS
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int kb0 = a.n_in >> 4, t0n = (kb0 + 1) >> 1;

    // a.img: the fp32 image (biases are read from it); a.img2: the split image
    auto dma_layer = [&](int layer) {
        const int npiece = ams_steps(a.n_in, layer) * ams_ibn(layer) * 2;
        const float* src = a.img2 + ams_layer_off(a.n_in, layer);
        float* dst = smem + (layer & 1) * AM_BUF_FLOATS;
        for (int q = wave; q < npiece; q += AMS_WAVES)
            __builtin_amdgcn_global_load_lds((rf_gptr)(src + q * 256 + lane * 4), (rf_lptr)(dst + q * 256), 16, 0, 0);
    };
    const float4* bias4 = reinterpret_cast<const float4*>(a.img + am_layer_off(a.n_in, 4));     // [layer][32 float4s]

    dma_layer(0);
    const int nwt = (a.ntiles + AMS_WAVES - 1) / AMS_WAVES;             // workgroup tiles of 12 waves x 16 rows
    for (int wt = blockIdx.x; wt < nwt; wt += gridDim.x) {
        const int rt = wt * AMS_WAVES + wave;
        const bool live = rt < a.ntiles;
        int row = rt * 16 + j;
        if (row >= a.nrows) row = a.nrows - 1;                        // clamp: computed, never stored
        if (row < 0) row = 0;
        size_t orow = (size_t)row;
        am_h8 bh[4], bl[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { bh[t] = (am_h8){0, 0, 0, 0, 0, 0, 0, 0}; bl[t] = bh[t]; }
        if (a.mode == 0) {
            const float* p = a.src + (size_t)row * a.n_in + 4 * g;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                if (kb < kb0) {
                    const float4 v = *reinterpret_cast<const float4*>(p + kb * 16);
                    ams_split4((f32x4){v.x, v.y, v.z, v.w}, bh[kb >> 1], bl[kb >> 1], 4 * (kb & 1));
                }
            }
        } else {
            const int r = a.s >> 1, t = a.t, q = a.s / t;
            const int r3 = r * r * r;
            const int vol = row / r3, prow = row - vol * r3;
            const int p2 = prow % r, p1 = (prow / r) % r, p0 = prow / (r * r);
            const int bb = vol / a.kv, k = vol - bb * a.kv;
            orow = ((size_t)bb * r3 + prow) * a.kv + k;
            const int d0 = 2 * p0 + (g & 1), d1 = 2 * p1, d2 = 2 * p2;
            const size_t t3 = (size_t)t * t * t;
            const size_t patch = (((size_t)vol * q + d0 / t) * q + d1 / t) * q + d2 / t;
            const float* p = a.src + (patch * a.c + (g >> 1)) * t3 + ((size_t)(d0 % t) * t + (d1 % t)) * t + (d2 % t);
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                if (kb < kb0) {                                        // features 16kb+4g+r: channel 2kb+(g>>1), e0 = g&1, (e1,e2) = r
                    const float2 lo2 = *reinterpret_cast<const float2*>(p + (size_t)(2 * kb) * t3);
                    const float2 hi2 = *reinterpret_cast<const float2*>(p + (size_t)(2 * kb) * t3 + t);
                    ams_split4((f32x4){lo2.x, lo2.y, hi2.x, hi2.y}, bh[kb >> 1], bl[kb >> 1], 4 * (kb & 1));
                }
            }
        }

        f32x4 hi[8], lo[8];
#pragma unroll
        for (int layer = 0; layer < 4; ++layer) {
            __syncthreads();                                          // weights of `layer` landed; everyone left layer-1's buffer
            if (layer < 3) dma_layer(layer + 1);
            else if (wt + (int)gridDim.x < nwt) dma_layer(0);
            const float* wbuf = smem + (layer & 1) * AM_BUF_FLOATS;
            if (layer < 3) {
                ams_layer<8>(wbuf, layer == 0 ? t0n : 4, bh, bl, hi, lo, lane);
#pragma unroll
                for (int ib = 0; ib < 8; ++ib) {                       // bias + LeakyReLU(0.01) + split: the next layer's B operands
                    const float4 bz = bias4[layer * 32 + ib * 4 + g];
                    f32x4 v;
                    v[0] = fmaf(lo[ib][0], 1.0f / AMS_LO, hi[ib][0]) + bz.x; v[1] = fmaf(lo[ib][1], 1.0f / AMS_LO, hi[ib][1]) + bz.y;
                    v[2] = fmaf(lo[ib][2], 1.0f / AMS_LO, hi[ib][2]) + bz.z; v[3] = fmaf(lo[ib][3], 1.0f / AMS_LO, hi[ib][3]) + bz.w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * 0.01f;
                    ams_split4(v, bh[ib >> 1], bl[ib >> 1], 4 * (ib & 1));
                }
            } else {
                ams_layer<2>(wbuf, 4, bh, bl, hi, lo, lane);
                if (live && rt * 16 + j < a.nrows) {
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) {
                        const float4 bz = bias4[3 * 32 + ib * 4 + g];
                        const float4 o = make_float4(fmaf(lo[ib][0], 1.0f / AMS_LO, hi[ib][0]) + bz.x, fmaf(lo[ib][1], 1.0f / AMS_LO, hi[ib][1]) + bz.y,
                                                     fmaf(lo[ib][2], 1.0f / AMS_LO, hi[ib][2]) + bz.z, fmaf(lo[ib][3], 1.0f / AMS_LO, hi[ib][3]) + bz.w);
                        *reinterpret_cast<float4*>(a.out + orow * AM_OUT + ib * 16 + 4 * g) = o;
                    }
                }
            }
        }
    }
}

static int am_launch(const AmArgs& a, hipStream_t st) {
    const int lds = 2 * AM_BUF_FLOATS * (int)sizeof(float);
    static RfLdsOptIn opt_in;
    if (int rc = opt_in.ensure(reinterpret_cast<const void*>(k_attn_mlp), lds, "rf_attn_mlp")) return rc;
    const int nwt = (a.ntiles + AM_WAVES - 1) / AM_WAVES;
    if (a.img2) {
        static RfLdsOptIn opt_in2;
        if (int rc = opt_in2.ensure(reinterpret_cast<const void*>(k_attn_mlp_split), lds, "rf_attn_mlp")) return rc;
        const int nwt2 = (a.ntiles + AMS_WAVES - 1) / AMS_WAVES;
        hipLaunchKernelGGL(k_attn_mlp_split, dim3(nwt2 < 256 ? nwt2 : 256), dim3(AMS_WAVES * 64), lds, st, a);
    } else {
        hipLaunchKernelGGL(k_attn_mlp, dim3(nwt < 256 ? nwt : 256), dim3(AM_WAVES * 64), lds, st, a);
    }
    RF_CHECK_LAUNCH("rf_attn_mlp");
    return RF_OK;
}

static int am_rows(const float* x, int rows, int n_in, const float* packed, const float* packed_split, float* out, void* stream);
extern "C" int rf_attn_mlp_rows(const float* x, int rows, int n_in, const float* packed, float* out, void* stream) {
    return am_rows(x, rows, n_in, packed, nullptr, out, stream);
}
extern "C" int rf_attn_mlp_split_rows(const float* x, int rows, int n_in, const float* packed, const float* packed_split, float* out, void* stream) {
    RF_REQUIRE(packed_split, RF_E_INVALID, "rf_attn_mlp_split_rows: null split image");
    return am_rows(x, rows, n_in, packed, packed_split, out, stream);
}
static int am_rows(const float* x, int rows, int n_in, const float* packed, const float* packed_split, float* out, void* stream) {
    RF_REQUIRE(x && packed && out && rows > 0, RF_E_INVALID, "rf_attn_mlp_rows: bad arguments");
    RF_REQUIRE(n_in >= 16 && n_in <= AM_HID && n_in % 16 == 0, RF_E_UNSUPPORTED, "rf_attn_mlp_rows: n_in %d must be a multiple of 16 in 16..128", n_in);
    AmArgs a;
    a.src = x; a.img = packed; a.img2 = packed_split; a.out = out; a.mode = 0; a.nrows = rows; a.n_in = n_in; a.ntiles = (rows + 15) / 16;
    a.kv = 1; a.c = 0; a.s = 0; a.t = 0;
    return am_launch(a, (hipStream_t)stream);
}

static int am_volume(const float* src, int b, int kv, int c, int s, int t, const float* packed, const float* packed_split, float* out, void* stream);
extern "C" int rf_attn_mlp_volume(const float* src, int b, int kv, int c, int s, int t, const float* packed, float* out, void* stream) {
    return am_volume(src, b, kv, c, s, t, packed, nullptr, out, stream);
}
extern "C" int rf_attn_mlp_split_volume(const float* src, int b, int kv, int c, int s, int t, const float* packed, const float* packed_split, float* out, void* stream) {
    RF_REQUIRE(packed_split, RF_E_INVALID, "rf_attn_mlp_split_volume: null split image");
    return am_volume(src, b, kv, c, s, t, packed, packed_split, out, stream);
}
static int am_volume(const float* src, int b, int kv, int c, int s, int t, const float* packed, const float* packed_split, float* out, void* stream) {
    RF_REQUIRE(src && packed && out && b > 0 && kv > 0 && c > 0 && s > 0 && t > 0, RF_E_INVALID, "rf_attn_mlp_volume: bad arguments");
    RF_REQUIRE(s % 2 == 0 && t % 2 == 0 && s % t == 0, RF_E_INVALID, "rf_attn_mlp_volume: edges s=%d t=%d must be even and t | s", s, t);
    RF_REQUIRE(c % 2 == 0 && c * 8 <= AM_HID, RF_E_UNSUPPORTED, "rf_attn_mlp_volume: %d channels (need an even count <= 16)", c);
    const long long rows = (long long)b * kv * (s / 2) * (s / 2) * (s / 2);
    RF_REQUIRE(rows < (1ll << 31) - 16, RF_E_UNSUPPORTED, "rf_attn_mlp_volume: too many rows");
    AmArgs a;
    a.src = src; a.img = packed; a.img2 = packed_split; a.out = out; a.mode = 1; a.nrows = (int)rows; a.n_in = c * 8; a.ntiles = (int)((rows + 15) / 16);
    a.kv = kv; a.c = c; a.s = s; a.t = t;
    return am_launch(a, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ weights per row
// Gumbel(0, 1) noise for the hard attention (reference model/attention.py:100-103: -log(Exponential(1)) per (row, k), drawn by torch's
// exponential_ from the global generator): Philox4x32-10 keyed by a 64-bit seed, counter = (row, draw index, offset), four draws per call;
// u = (24 random bits + 0.5) / 2^24 in (0, 1), g = -log(-log(u)).  The stream is this library's own (torch's generator cannot be advanced from
// inside a kernel); what the reference fixes is the DISTRIBUTION, which tests/test_kernels_gpu.py checks against torch's sampler.
__device__ __forceinline__ void rf_philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (unsigned)p1; c[3] = (unsigned)p0; c[0] = n0; c[2] = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ void rf_gumbel_row(unsigned long long seed, unsigned long long offset, int row, int K, float (&g)[RF_MAX_K]) {
#pragma unroll
    for (int q = 0; q < RF_MAX_K / 4; ++q) {
        if (4 * q < K) {
            unsigned c[4] = {(unsigned)row, (unsigned)q, (unsigned)offset, (unsigned)(offset >> 32)};
            rf_philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float u = ((float)(c[j] >> 8) + 0.5f) * (1.0f / 16777216.0f);
                g[4 * q + j] = -logf(-logf(u));
            }
        }
    }
}

// one THREAD per attention row (a row is 32 + K*32 floats of features: nothing to share across lanes); the arithmetic is
// rf_attn_row_weights, shared with k_attn_fuse.  rng (Gumbel-hard mode without a noise tensor): {seed, offset, finished-blocks counter}; the
// last block to finish advances the offset, so a captured graph draws fresh noise at every replay.
__global__ __launch_bounds__(256) void k_attn_weights(const float* __restrict__ xf, const float* __restrict__ pf, const float* __restrict__ noise,
                                                      unsigned long long* __restrict__ rng, int rows, int K, int f, int mode, float sharpness,
                                                      float* __restrict__ w_out, float* __restrict__ sw_out, float* __restrict__ scores_out,
                                                      float* __restrict__ noise_out) {
    const unsigned long long seed = rng ? rng[0] : 0ull, offset = rng ? rng[1] : 0ull;
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += gridDim.x * blockDim.x) {
        float sc[RF_MAX_K], w[RF_MAX_K], sw, g[RF_MAX_K];
        const float* nrow = noise ? noise + (size_t)row * K : nullptr;
        if (!noise && mode == RF_ATTN_GUMBEL_HARD) {
            rf_gumbel_row(seed, offset, row, K, g);
            nrow = g;
        }
        rf_attn_row_weights(xf + (size_t)row * f, pf + (size_t)row * K * f, nrow, K, f, mode, sharpness, sc, w, sw);
#pragma unroll
        for (int k = 0; k < RF_MAX_K; ++k) {
            if (k < K) {
                w_out[(size_t)row * K + k] = w[k];
                if (scores_out) scores_out[(size_t)row * K + k] = sc[k];
                if (noise_out && nrow) noise_out[(size_t)row * K + k] = nrow[k];
            }
        }
        sw_out[row] = sw;
    }
    if (rng && !noise && mode == RF_ATTN_GUMBEL_HARD) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            if (atomicAdd(rng + 2, 1ull) == (unsigned long long)gridDim.x - 1ull) { rng[1] = offset + 1ull; rng[2] = 0ull; }
        }
    }
}

static int attn_weights_launch(const float* xf, const float* pf, const float* noise, unsigned long long* rng, int rows, int k, int f, int mode,
                               float sharpness, float* weights, float* switches, float* scores_out, float* noise_out, void* stream, const char* who) {
    RF_REQUIRE(xf && pf && weights && switches && rows > 0, RF_E_INVALID, "%s: bad arguments", who);
    RF_REQUIRE(k >= 1 && k <= RF_MAX_K, RF_E_UNSUPPORTED, "%s: K=%d outside 1..%d", who, k, RF_MAX_K);
    RF_REQUIRE(f >= 1 && f <= 128, RF_E_UNSUPPORTED, "%s: feature width %d outside 1..128", who, f);
    const int want = (rows + 255) / 256;
    hipLaunchKernelGGL(k_attn_weights, dim3(want < 8192 ? want : 8192), dim3(256), 0, (hipStream_t)stream, xf, pf, noise, rng, rows, k, f, mode, sharpness,
                       weights, switches, scores_out, noise_out);
    RF_CHECK_LAUNCH(who);
    return RF_OK;
}

extern "C" int rf_attn_weights(const float* xf, const float* pf, const float* noise, int rows, int k, int f, int mode, float sharpness,
                               float* weights, float* switches, float* scores_out, void* stream) {
    RF_REQUIRE(mode == RF_ATTN_SOFTMAX || (mode == RF_ATTN_GUMBEL_HARD && noise), RF_E_INVALID,
               "rf_attn_weights: Gumbel-hard mode needs the noise tensor (or rf_attn_weights_sampled)");
    return attn_weights_launch(xf, pf, noise, nullptr, rows, k, f, mode, sharpness, weights, switches, scores_out, nullptr, stream, "rf_attn_weights");
}

// Gumbel-hard weights with the noise drawn inside the kernel (see rf_gumbel_row).  rng_state: 3 x uint64 on the device = {seed, offset, 0};
// every call advances the offset.  noise_out (optional) receives the noise that was used: feeding it to rf_attn_weights reproduces the weights.
extern "C" int rf_attn_weights_sampled(const float* xf, const float* pf, int rows, int k, int f, float sharpness, void* rng_state,
                                       float* weights, float* switches, float* scores_out, float* noise_out, void* stream) {
    RF_REQUIRE(rng_state, RF_E_INVALID, "rf_attn_weights_sampled: null generator state");
    return attn_weights_launch(xf, pf, nullptr, reinterpret_cast<unsigned long long*>(rng_state), rows, k, f, RF_ATTN_GUMBEL_HARD, sharpness, weights,
                               switches, scores_out, noise_out, stream, "rf_attn_weights_sampled");
}

// ------------------------------------------------------------------------------------------------ blend in the folded layout
// one thread per voxel pair along x (= one attention patch row of extent 2): out = x*(1-sw) + (sum_k w_k p_k)*sw
__global__ __launch_bounds__(256) void k_attn_blend(const float* __restrict__ x, const float* __restrict__ src, const float* __restrict__ w,
                                                    const float* __restrict__ sw, int b, int K, int c, int s, int t, float* __restrict__ out) {
    const int r = s >> 1, q = s / t;
    const size_t t3 = (size_t)t * t * t, total = (size_t)b * c * s * s * r;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p2 = (int)(i % r), d1 = (int)((i / r) % s), d0 = (int)((i / ((size_t)r * s)) % s);
        const int cc = (int)((i / ((size_t)r * s * s)) % c);
        const size_t bb = i / ((size_t)r * s * s * c);
        const size_t row = ((bb * r + (d0 >> 1)) * r + (d1 >> 1)) * r + p2;
        const int d2 = 2 * p2;
        const float swv = sw[row];
        float ws0 = 0.f, ws1 = 0.f;
        for (int k = 0; k < K; ++k) {
            const size_t vol = bb * K + k;
            const size_t patch = ((vol * q + d0 / t) * q + d1 / t) * q + d2 / t;
            const float2 pv = *reinterpret_cast<const float2*>(src + (patch * c + cc) * t3 + ((size_t)(d0 % t) * t + (d1 % t)) * t + (d2 % t));
            const float wk = w[row * K + k];
            ws0 = fmaf(wk, pv.x, ws0);
            ws1 = fmaf(wk, pv.y, ws1);
        }
        const float2 xv = reinterpret_cast<const float2*>(x)[i];
        // torch: x*(1-s) + p*s with every product rounded (no FMA contraction), as k_attn_fuse
        reinterpret_cast<float2*>(out)[i] = make_float2(__fadd_rn(__fmul_rn(xv.x, 1.f - swv), __fmul_rn(ws0, swv)),
                                                        __fadd_rn(__fmul_rn(xv.y, 1.f - swv), __fmul_rn(ws1, swv)));
    }
}

extern "C" int rf_attn_blend(const float* x, const float* retrieved, int b, int k, int c, int s, int t, const float* weights,
                             const float* switches, float* out, void* stream) {
    RF_REQUIRE(x && retrieved && weights && switches && out && b > 0 && k > 0 && c > 0, RF_E_INVALID, "rf_attn_blend: bad arguments");
    RF_REQUIRE(s > 0 && t > 0 && s % 2 == 0 && t % 2 == 0 && s % t == 0, RF_E_INVALID, "rf_attn_blend: edges s=%d t=%d must be even and t | s", s, t);
    const size_t total = (size_t)b * c * s * s * (s / 2);
    const size_t want = (total + 255) / 256;
    hipLaunchKernelGGL(k_attn_blend, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, (hipStream_t)stream, x, retrieved, weights,
                       switches, b, k, c, s, t, out);
    RF_CHECK_LAUNCH("rf_attn_blend");
    return RF_OK;
}
