// rf_conv3d_valid_leaky_split: valid (no padding) strided Conv3d + bias + LeakyReLU of the conv patch encoders (reference
// model/retrieval.py:4-28 Patch32, :217-243 PCPatch48; the layers rf_conv3d_valid_leaky_lds takes) on the F16 matrix cores by
// OPERAND SPLITTING (conv3d_up_split.hip has the numerics):
//     x = h + l / 2^11,  h = f16(x),  l = f16((x - h) * 2^11);   a*b ~ ah*bh + (ah*bl + al*bh) / 2^11,   exact f16 x f16 products,
//     fp32 accumulation in two accumulators (hi, lo), combined once in the epilogue.
//
// GEMM view: M = output voxels of a tile (tz x ty x tx voxels of one volume -- whole output rows for the encoders' windows, x segments for
// the big grids of the fully-convolutional evaluation -- linear index, 4 waves x 4 m-blocks = 256), N = cout
// (NB <= 3 n-blocks per workgroup), K = k^3 * cin walked in PIECES: a piece is (tap, group of 4 input channels); an MFMA k-step (k = 32)
// is 8 pieces, lane group g = lane >> 4 supplies pieces 8q + 2g and 8q + 2g + 1.  The encoders' channel counts are multiples of 4, not
// of 8 (PCPatch48: 12, 24, 48, 96), and with 4-channel pieces in flat order a 12-channel layer needs 11 k-steps where 8-channel slots
// would need 14.  LDS image of a chunk of `cgc` channel groups: [group][input position] in 8-byte slots (4 channels of one input voxel),
// one plane for h, one for l; positions are the tile's (tz-1)*stride + k input planes of (ty-1)*stride + k input rows of (tx-1)*stride + k voxels.  An A
// operand is four ds_read_b64 at (output corner + piece offset), the piece offsets of a k-step come from a table in LDS.
// Staging: 4 channel planes -> one item (position, group) per thread step, scaled, clamped and split on the way in; no LDS double
// buffer -- two or three workgroups per CU overlap one's staging with the other's MFMAs.
// B operands (weights): f16 fragment image from rf_convv_split_pack_weight ([chunk][k-step][n-block][h|l][lane][8 halves]), L2-resident,
// global -> VGPR one k-step ahead (across chunk boundaries too).
#include "common.h"
#include <type_traits>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

namespace {
constexpr float VS_ACT_SCALE = 1.0f / 16, VS_W_SCALE = 16.0f, VS_LO = 2048.0f;
constexpr int VS_NT = 256, VS_MB = 4, VS_M = 256;               // threads, m-blocks per wave, output voxels per tile
constexpr int VS_EV = VS_M + 4;                                 // floats per cout row of the epilogue tile
constexpr size_t VS_LDS_MAX = 78 * 1024;                        // two workgroups per CU
constexpr int VS_SB = 11;                                       // staging items (4 channels each) in flight per thread
}   // namespace

struct ConvVSArgs {
    const float* x;
    const h8* wp;
    const float* bias;
    float* out;
    int n, cin, s, cout, k, stride, so;
    float slope;
    int tz, ty, tx, ntz, nty, ntx, gz;   // output voxels per tile and dim, tiles per dim, groups of NB cout blocks
    int zi, yi, xi, npos;          // staged input planes / rows per plane / voxels per row, positions per channel group (zi * yi * xi)
    int cgc, nchunk, ksteps;       // 4-channel groups per chunk, chunks, k-steps per chunk
    int nbt;                       // n-blocks of the weight image (cout16 / 16)
    int out_pre;                   // 1: the output is written in split form (rf_valid_split_act: [4-channel group][h | l][voxel][4 halves])
    int hdr;                       // ints of the image's table header, a multiple of 4
};

// tile and chunk choice; depends on the layer only (not on n): the weight image is packed for it
static bool convv_split_plan(int cin, int s, int cout, int k, int stride, ConvVSArgs& a, int& nb_out, size_t& lds_out) {
    if (cin <= 0 || cout <= 0 || k < 2 || k > 5 || stride < 1 || stride > 2 || s < k || s > 255 || (cin & 3)) return false;
    const int so = (s - k) / stride + 1;
    if (so < 6) return false;
    const int cout16 = rf_round_up(cout, 16);
    const int nbw = cout16 <= 48 ? cout16 / 16 : (cout16 % 48 == 0 ? 3 : 2);
    const int k3 = k * k * k, cgt = cin / 4;
    double best = 0.0;
    a.npos = 0; a.tx = 0;
    for (int tz = 1; tz <= so && tz <= VS_M; ++tz)
        for (int ty = 1; ty <= so && tz * ty <= VS_M; ++ty)
            for (int tx = so; tx >= 4; --tx) {
                // whole rows when they fit (the windows of the patch encoders); segments of at least 16 voxels (64-byte runs) otherwise
                if (tx != so && (so <= VS_M / 4 || tx < 16 || (tx & 3))) continue;
                const int V = tz * ty * tx;
                if (V > VS_M) continue;
                const int zi = (tz - 1) * stride + k, yi = (ty - 1) * stride + k, xi = (tx - 1) * stride + k;
                const int npos = zi * yi * xi;
                const int ntz = (so + tz - 1) / tz, nty = (so + ty - 1) / ty, ntx = (so + tx - 1) / tx;
                const double tile_eff = (double)so * so * so / ((double)ntz * nty * ntx * VS_M);
                for (int cgc = 1; cgc <= cgt; ++cgc) {
                    if (cgt % cgc) continue;
                    const int ksteps = (k3 * cgc + 7) / 8;
                    if (ksteps > 64) continue;                          // piece table: two entries per thread
                    const int items_pad = rf_round_up(cgc * npos, VS_SB * VS_NT);       // the staging loop writes whole batches
                    size_t lds = (size_t)2 * items_pad * 8 + (size_t)ksteps * 8 * 4 + (size_t)(items_pad / xi + 2) * 4;
                    if (lds > VS_LDS_MAX) continue;
                    if (lds < (size_t)16 * VS_EV * 4) lds = (size_t)16 * VS_EV * 4;     // the epilogue tile aliases the image
                    const double eff = tile_eff * (double)(k3 * cgc) / (8.0 * ksteps);
                    if (eff > best + 1e-9 || (eff > best - 1e-9 && npos < a.npos)) {    // ties: the smaller staged tile
                        best = eff;
                        a.tz = tz; a.ty = ty; a.tx = tx; a.ntz = ntz; a.nty = nty; a.ntx = ntx; a.zi = zi; a.yi = yi; a.xi = xi; a.npos = npos;
                        a.cgc = cgc; a.nchunk = cgt / cgc; a.ksteps = ksteps;
                        lds_out = lds;
                    }
                }
            }
    if (best < 0.5) return false;
    a.cin = cin; a.s = s; a.cout = cout; a.k = k; a.stride = stride; a.so = so;
    a.nbt = cout16 / 16;
    a.gz = (a.nbt + nbw - 1) / nbw;
    a.hdr = rf_round_up(a.ksteps * 8 + 3 * VS_M, 4);
    nb_out = nbw;
    return true;
}

// ------------------------------------------------------------------------------------------------------------ weight image
extern "C" size_t rf_convv_split_packed_bytes(int cout, int cin, int k, int s, int stride) {
    ConvVSArgs a;
    int nb;
    size_t lds;
    if (!convv_split_plan(cin, s, cout, k, stride, a, nb, lds)) return 0;
    // two k-steps of slack: the kernel's one-ahead B prefetch (NB blocks from the last group's first block) never leaves the image
    return (size_t)a.hdr * 4 + ((size_t)a.nchunk * a.ksteps + 2) * (size_t)a.nbt * 2 * 64 * 16;
}

// image header: the tables every workgroup needs and that depend on the layer only --
//   [ksteps * 8] byte offset of piece p in the h plane (zero-weight pad pieces: 0, a position that is always staged);
//   [VS_M]       byte offset of the input corner of tile voxel m in a plane (m >= tile size: 0 -- computed, never stored);
//   [VS_M]       offset of tile voxel m in the output window relative to the tile's first voxel (m >= tile size: -1);
//   [VS_M]       (lz << 16) | (ly << 8) | lx of tile voxel m (ragged last tiles)
__global__ void k_convv_split_header(ConvVSArgs a, int* __restrict__ hdr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int np = a.ksteps * 8;
    if (i < np) {
        const int k = a.k;
        int off = 0;
        if (i < k * k * k * a.cgc) {
            const int tap = i / a.cgc, cg = i - tap * a.cgc;
            off = (cg * a.npos + ((tap / (k * k)) * a.yi + (tap / k) % k) * a.xi + tap % k) * 8;
        }
        hdr[i] = off;
    } else if (i < np + 3 * VS_M) {
        const int which = (i - np) / VS_M, m0 = (i - np) % VS_M;
        const bool in = m0 < a.tz * a.ty * a.tx;
        const int m = in ? m0 : 0;
        const int x = m % a.tx, r = m / a.tx, ly = r % a.ty, lz = r / a.ty;
        if (which == 0) hdr[i] = (((lz * a.stride) * a.yi + ly * a.stride) * a.xi + x * a.stride) * 8;
        else if (which == 1) hdr[i] = in ? (lz * a.so + ly) * a.so + x : -1;
        else hdr[i] = (lz << 16) | (ly << 8) | x;
    } else if (i < a.hdr) {
        hdr[i] = 0;
    }
}

__global__ void k_convv_split_pack(const float* __restrict__ w, int cout, int cin, int k3, int cgc, int ksteps, int nbt, size_t nreal,
                                   h8* __restrict__ wp, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), part = (int)((i >> 6) & 1);
        const size_t f = i >> 7;
        const int nb = (int)(f % nbt);
        const size_t st = f / nbt;
        const int q = (int)(st % ksteps), chunk = (int)(st / ksteps);
        const int co = nb * 16 + (lane & 15), g = lane >> 4;
        h8 out;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = 8 * q + 2 * g + (j >> 2);
            double v = 0.0;
            if (i < nreal && p < k3 * cgc && co < cout) {
                const int tap = p / cgc, ci = (chunk * cgc + p % cgc) * 4 + (j & 3);
                v = (double)w[((size_t)co * cin + ci) * k3 + tap];
            }
            v *= (double)VS_W_SCALE;
            v = v > 65504.0 ? 65504.0 : (v < -65504.0 ? -65504.0 : v);
            const _Float16 h = (_Float16)(float)v;
            out[j] = part == 0 ? h : (_Float16)(float)((v - (double)(float)h) * (double)VS_LO);
        }
        wp[i] = out;
    }
}

extern "C" int rf_convv_split_pack_weight(const float* w_oidhw, int cout, int cin, int k, int s, int stride, void* w_packed, void* stream) {
    RF_REQUIRE(w_oidhw && w_packed, RF_E_INVALID, "rf_convv_split_pack_weight: null pointer");
    ConvVSArgs a;
    int nb;
    size_t lds;
    RF_REQUIRE(convv_split_plan(cin, s, cout, k, stride, a, nb, lds), RF_E_UNSUPPORTED,
               "rf_convv_split_pack_weight: layer not taken by the split form (ask rf_conv3d_valid_split_supported)");
    const size_t total = (rf_convv_split_packed_bytes(cout, cin, k, s, stride) - (size_t)a.hdr * 4) / 16;
    hipLaunchKernelGGL(k_convv_split_header, dim3((unsigned)((a.hdr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, reinterpret_cast<int*>(w_packed));
    const size_t nreal = (size_t)a.nchunk * a.ksteps * a.nbt * 128;
    const size_t want = (total + 255) / 256;
    hipLaunchKernelGGL(k_convv_split_pack, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream, w_oidhw, cout, cin,
                       k * k * k, a.cgc, a.ksteps, a.nbt, nreal, reinterpret_cast<h8*>(w_packed) + a.hdr / 4, total);
    RF_CHECK_LAUNCH("rf_convv_split_pack_weight");
    return RF_OK;
}

// ------------------------------------------------------------------------------------------------------------------- kernel
// On gfx9 loads and stores retire through ONE in-order counter (vmcnt): a wait for any load that was requested after a store also waits
// for that store's acknowledgement from L2.  The epilogue therefore requests nothing -- bias values and the store tables are loaded before
// the MFMAs, nothing may spill (a reload is a scratch load) -- and its stores stay in flight while the workgroup retires.
// PRE: the input is in split form already (written by the previous layer: rf_conv3d_valid_leaky_valu_to_split / ..._split_to_split) -- its 8-byte
// slots are the LDS image's slots, staging is a copy: no scaling, no conversion (47 % of this kernel's instruction issue was that VALU work)
template <int NB, int WPE, bool PRE>
__global__ __launch_bounds__(VS_NT, WPE) void k_convv_split(ConvVSArgs a) {
    constexpr int NT = VS_NT, MB = VS_MB, SB = VS_SB;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int so = a.so, s = a.s, st = a.stride, xi = a.xi;
    const int items = a.cgc * a.npos;
    const int items_pad = (items + SB * NT - 1) / (SB * NT) * (SB * NT);    // the staging loop writes whole batches (pad slots: never read)
    const int plane = items_pad * 8;                                // bytes of the h plane (l plane follows)
    int* poff = reinterpret_cast<int*>(lds + 2 * plane);            // [ksteps * 8] byte offset of a piece in the h plane
    int* rowsrc = poff + a.ksteps * 8;                              // [rows_pad] float offset of a staged row (< 0: outside the volume / the chunk)
    const int nrows = a.cgc * a.zi * a.yi, rows_pad = items_pad / xi + 2;
    const size_t ivol = (size_t)s * s * s;

    // XCD-aware 1-D grid as in k_convv_lds: an XCD walks whole windows (tiles fastest, then cout block groups)
    const unsigned total = gridDim.x, per = total >> 3, rem = total & 7u, xk = blockIdx.x & 7u;
    const unsigned lb = xk * per + (xk < rem ? xk : rem) + (blockIdx.x >> 3);
    const unsigned tiles = (unsigned)(a.ntz * a.nty * a.ntx);
    const unsigned tb = lb % tiles, zb = (lb / tiles) % (unsigned)a.gz;
    const int nn = (int)(lb / (tiles * (unsigned)a.gz));
    const unsigned tzy = tb / (unsigned)a.ntx;
    const int z0 = (int)(tzy / (unsigned)a.nty) * a.tz, y0 = (int)(tzy % (unsigned)a.nty) * a.ty, x0t = (int)(tb % (unsigned)a.ntx) * a.tx;
    const int xin0 = x0t * st, xlast = s - 1 - xin0;              // first staged input column; the last column of the row relative to it
    const int nb0 = (int)zb * NB;                                   // first n-block of this workgroup

    // tables from the image header; the staged rows of this tile
    const int* hdr = reinterpret_cast<const int*>(a.wp);
    const bool ragged = z0 * st + a.zi > s || y0 * st + a.yi > s;   // uniform
    int pv[2];                                                      // piece offsets: requested now, written to LDS behind the staging loop
#pragma unroll
    for (int h = 0; h < 2; ++h) pv[h] = tid + h * NT < a.ksteps * 8 ? hdr[tid + h * NT] : 0;
    {
        const int rows_g = a.zi * a.yi;
        const float inv_g = 1.0f / (float)rows_g, inv_y = 1.0f / (float)a.yi;     // r < 2^16: (r + 0.5) * (1 / d) truncates to r / d
        for (int r = tid; r < rows_pad; r += NT) {
            const int cg = (int)(((float)r + 0.5f) * inv_g), rr = r - cg * rows_g;
            const int rz = (int)(((float)rr + 0.5f) * inv_y), ry = rr - rz * a.yi;
            const int iz = z0 * st + rz, iy = y0 * st + ry;
            rowsrc[r] = r >= nrows ? 0 : (iz < s && iy < s) ? (int)((size_t)cg * (PRE ? 2 : 4) * ivol) + (iz * s + iy) * s + xin0 : -1;   // ragged last tile: rows past the volume
        }
    }
    int base[MB];                                                   // byte offset of the input corner of voxel (m-block, j) in a plane
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) base[mb] = hdr[a.ksteps * 8 + (wave * MB + mb) * 16 + j];
    int eoff[4], ezy[4];                                            // epilogue: voxel m = lane + 64 i -> offset in the output volume, (lz << 16) | (ly << 8) | lx
#pragma unroll
    for (int i = 0; i < 4; ++i) { eoff[i] = hdr[a.ksteps * 8 + VS_M + lane + 64 * i]; ezy[i] = hdr[a.ksteps * 8 + 2 * VS_M + lane + 64 * i]; }
    float bz[NB];                                                   // before any store (see above)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int co = (nb0 + nb) * 16 + j;
        bz[nb] = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
    }

    const int step_r = NT / xi, step_x = NT - step_r * xi;          // item index advances by NT: (row, x) += (step_r, step_x) with carry
    const int row0 = tid / xi, x0 = tid - row0 * xi;

    f32x4 hi[MB][NB], lo[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { hi[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; lo[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const h8* wl = a.wp + a.hdr / 4 + (size_t)nb0 * 128 + lane;
    const size_t wstep = (size_t)a.nbt * 128;                       // h8 per k-step of the image
    h8 bh[NB], bl[NB], nh[NB], nl[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { bh[nb] = wl[nb * 128]; bl[nb] = wl[nb * 128 + 64]; }

    for (int c = 0; c < a.nchunk; ++c) {
        __syncthreads();                                           // tables written / everyone left the previous chunk
        {
            // SB items (4 channels each) in flight per thread, no branches around the loads.  Rows of a ragged last tile that lie outside
            // the volume (table entry < 0) read the channel's first value and are zeroed by their scale; tiles without such rows (nearly
            // all) take the path without that bookkeeping.  Pad slots behind the last item hold whatever the first row holds: never read.
            const char* xc = reinterpret_cast<const char*>(a.x + ((size_t)nn * a.cin + (size_t)c * a.cgc * 4) * ivol);       // (same byte offset in either form)
            auto stage = [&](auto ragged_tag) {
                constexpr bool RAGGED = decltype(ragged_tag)::value;
                int row = row0, ix = x0;
                for (int i = tid; i < items; i += SB * NT) {
                    float v[PRE ? 1 : SB][4];
                    uint2 ph[PRE ? SB : 1], pl[PRE ? SB : 1];
                    unsigned off[SB];
                    unsigned real = 0;
#pragma unroll
                    for (int b = 0; b < SB; ++b) {
                        const int ro = rowsrc[row];
                        const int ixc = ix < xlast ? ix : xlast;       // columns past the row (last x tile): some value of the row, only unstored voxels see it
                        if constexpr (RAGGED) {
                            off[b] = ro >= 0 ? (unsigned)(ro + ixc) * (PRE ? 8u : 4u) : 0u;
                            real |= (ro >= 0 ? 1u : 0u) << b;
                        } else {
                            off[b] = (unsigned)(ro + ixc) * (PRE ? 8u : 4u);
                        }
                        row += step_r; ix += step_x;
                        if (ix >= xi) { ix -= xi; ++row; }
                    }
                    if constexpr (PRE) {
#pragma unroll
                        for (int b = 0; b < SB; ++b) {
                            ph[b] = *reinterpret_cast<const uint2*>(xc + off[b]);
                            pl[b] = *reinterpret_cast<const uint2*>(xc + (size_t)8 * ivol + off[b]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int b = 0; b < SB; ++b) {
                            const bool keep = !RAGGED || ((real >> b) & 1u);
                            const int idx = i + b * NT;
                            *reinterpret_cast<uint2*>(lds + idx * 8) = keep ? ph[b] : make_uint2(0u, 0u);
                            *reinterpret_cast<uint2*>(lds + plane + idx * 8) = keep ? pl[b] : make_uint2(0u, 0u);
                        }
                    } else {
#pragma unroll
                        for (int b = 0; b < SB; ++b)
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[b][e] = *reinterpret_cast<const float*>(xc + (size_t)e * 4 * ivol + off[b]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int b = 0; b < SB; ++b) {
                            const float sc = (!RAGGED || ((real >> b) & 1u)) ? VS_ACT_SCALE : 0.f;
                            h4 hh, ll;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float t = __builtin_amdgcn_fmed3f(v[b][e] * sc, -65504.f, 65504.f);
                                const _Float16 h = (_Float16)t;
                                hh[e] = h;
                                ll[e] = (_Float16)fmaf(-VS_LO, (float)h, t * VS_LO);     // (t - h) * 2^11, exact either way; one v_fma_mix_f32
                            }
                            const int idx = i + b * NT;
                            *reinterpret_cast<h4*>(lds + idx * 8) = hh;
                            *reinterpret_cast<h4*>(lds + plane + idx * 8) = ll;
                        }
                    }
                }
            };
            if (ragged) stage(std::true_type{});
            else stage(std::false_type{});
        }
        if (c == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (tid + h * NT < a.ksteps * 8) poff[tid + h * NT] = pv[h];
        }
        __syncthreads();
        int2 po = *reinterpret_cast<const int2*>(poff + 2 * g);
        for (int q = 0; q < a.ksteps; ++q) {
            {   // next k-step's weights (the image has two k-steps of slack behind the last one)
                const h8* wn = wl + ((size_t)c * a.ksteps + q + 1) * wstep;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) { nh[nb] = wn[nb * 128]; nl[nb] = wn[nb * 128 + 64]; }
            }
            const int2 pn = *reinterpret_cast<const int2*>(poff + (q + 1 < a.ksteps ? q + 1 : q) * 8 + 2 * g);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const unsigned char* p0 = lds + base[mb] + po.x;
                const unsigned char* p1 = lds + base[mb] + po.y;
                const h4 a0 = *reinterpret_cast<const h4*>(p0), a1 = *reinterpret_cast<const h4*>(p1);
                const h4 c0 = *reinterpret_cast<const h4*>(p0 + plane), c1 = *reinterpret_cast<const h4*>(p1 + plane);
                const h8 ah = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                const h8 al = __builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) hi[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[nb], hi[mb][nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) lo[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[nb], lo[mb][nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) lo[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[nb], lo[mb][nb], 0, 0, 0);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) { bh[nb] = nh[nb]; bl[nb] = nl[nb]; }
            po = pn;
        }
    }
    __syncthreads();                                               // the chunk image is dead: the epilogue tile aliases it

    // ---- epilogue: hi + lo / 2^11 (activation and weight scales cancel), bias, LeakyReLU; through LDS so that the stores are long
    // contiguous runs -- per cout block the 4 waves each stream four cout rows out, lane = consecutive voxel of the tile
    static_assert(VS_ACT_SCALE * VS_W_SCALE == 1.0f, "epilogue assumes the operand scales cancel");
    float* eb = reinterpret_cast<float*>(lds);                      // [16][VS_EV]
    const size_t ovol = (size_t)so * so * so;
    int zlim = so - z0;
    if (zlim > a.tz) zlim = a.tz;
    int ylim = so - y0;
    if (ylim > a.ty) ylim = a.ty;
    int xlim = so - x0t;
    if (xlim > a.tx) xlim = a.tx;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (!((ezy[i] >> 16) < zlim && ((ezy[i] >> 8) & 255) < ylim && (ezy[i] & 255) < xlim)) eoff[i] = -1;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = fmaf(lo[mb][nb][r], 1.0f / VS_LO, hi[mb][nb][r]) + bz[nb];
                v[r] = t > 0.f ? t : t * a.slope;
            }
            *reinterpret_cast<f32x4*>(eb + j * VS_EV + (wave * MB + mb) * 16 + g * 4) = v;
        }
        __syncthreads();
        if (a.out_pre) {
            // split-form output for the next layer: this wave's four cout rows are ONE 4-channel group; lane = voxel, the group's h and l slots (8 bytes
            // each) leave as 512-byte runs.  Same scale / clamp / split as the consumer's staging would apply.
            const int grp = (nb0 + nb) * 4 + wave;
            if (grp * 4 < a.cout) {                                 // wave-uniform (cout is a multiple of 4)
                unsigned char* o = reinterpret_cast<unsigned char*>(a.out) + (((size_t)nn * (a.cout >> 2) + grp) * 2 * ovol + ((size_t)z0 * so + y0) * so + x0t) * 8;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (eoff[i] >= 0) {
                        h4 hh, ll;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = __builtin_amdgcn_fmed3f(eb[(wave * 4 + e) * VS_EV + lane + 64 * i] * VS_ACT_SCALE, -65504.f, 65504.f);
                            const _Float16 h = (_Float16)t;
                            hh[e] = h;
                            ll[e] = (_Float16)fmaf(-VS_LO, (float)h, t * VS_LO);
                        }
                        *reinterpret_cast<h4*>(o + (size_t)eoff[i] * 8) = hh;
                        *reinterpret_cast<h4*>(o + (ovol + (size_t)eoff[i]) * 8) = ll;
                    }
            }
        } else
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int col = wave * 4 + h, co = (nb0 + nb) * 16 + col;
            if (co < a.cout) {                                      // wave-uniform
                float* o = a.out + ((size_t)nn * a.cout + co) * ovol + ((size_t)z0 * so + y0) * so + x0t;
                const float* src = eb + col * VS_EV + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (eoff[i] >= 0) o[eoff[i]] = src[64 * i];
            }
        }
        if (nb + 1 < NB) __syncthreads();                          // the next block's writers wait for these readers; the stores stay in flight
    }
}

extern "C" int rf_conv3d_valid_split_supported(int n, int cin, int s, int cout, int k, int stride) {
    ConvVSArgs a;
    int nb;
    size_t lds;
    return n > 0 && convv_split_plan(cin, s, cout, k, stride, a, nb, lds) ? 1 : 0;
}

// x [n][cin][s^3] (in_split: in split form, rf_valid_split_act_bytes), w_packed from rf_convv_split_pack_weight for the same (cout, cin, k, s, stride),
// out [n][cout][so^3] (out_split: in split form; cout a multiple of 4)
extern "C" int rf_conv3d_valid_leaky_split_ex(const void* x, int in_split, int n, int cin, int s, const void* w_packed, const float* bias, int cout, int k,
                                              int stride, float slope, void* out, int out_split, void* stream) {
    RF_REQUIRE(x && w_packed && out && n > 0, RF_E_INVALID, "rf_conv3d_valid_leaky_split: bad arguments");
    RF_REQUIRE(!out_split || (cout & 3) == 0, RF_E_INVALID, "rf_conv3d_valid_leaky_split: a split-form output needs cout in multiples of 4 (got %d)", cout);
    ConvVSArgs a;
    int nbw;
    size_t lds;
    RF_REQUIRE(convv_split_plan(cin, s, cout, k, stride, a, nbw, lds), RF_E_UNSUPPORTED,
               "rf_conv3d_valid_leaky_split: layer not taken by the split form (ask rf_conv3d_valid_split_supported)");
    a.n = n; a.x = reinterpret_cast<const float*>(x); a.wp = reinterpret_cast<const h8*>(w_packed); a.bias = bias; a.out = reinterpret_cast<float*>(out); a.slope = slope;
    a.out_pre = out_split ? 1 : 0;
    const size_t grid64 = (size_t)a.ntz * a.nty * a.ntx * a.gz * n;
    RF_REQUIRE(grid64 < (1ull << 31), RF_E_INVALID, "rf_conv3d_valid_leaky_split: too many tiles (%zu)", grid64);
    const unsigned grid = (unsigned)grid64;
    hipStream_t st = (hipStream_t)stream;
#define RF_VS_LAUNCH(NB_, WPE_, PRE_)                                                                                            \
    do {                                                                                                                         \
        if (lds > 65536) {                                                                                                       \
            static RfLdsOptIn opt_in;                                                                                            \
            if (int rc = opt_in.ensure(reinterpret_cast<const void*>(k_convv_split<NB_, WPE_, PRE_>), (int)VS_LDS_MAX, "rf_conv3d_valid_leaky_split")) return rc; \
        }                                                                                                                        \
        hipLaunchKernelGGL((k_convv_split<NB_, WPE_, PRE_>), dim3(grid), dim3(VS_NT), lds, st, a);                               \
    } while (0)
    if (in_split) {
        switch (nbw) {
            case 1: RF_VS_LAUNCH(1, 3, true); break;
            case 2: RF_VS_LAUNCH(2, 3, true); break;
            default: RF_VS_LAUNCH(3, 2, true); break;
        }
    } else {
        switch (nbw) {
            case 1: RF_VS_LAUNCH(1, 3, false); break;
            case 2: RF_VS_LAUNCH(2, 3, false); break;
            default: RF_VS_LAUNCH(3, 2, false); break;
        }
    }
#undef RF_VS_LAUNCH
    RF_CHECK_LAUNCH("rf_conv3d_valid_leaky_split");
    return RF_OK;
}

extern "C" int rf_conv3d_valid_leaky_split(const float* x, int n, int cin, int s, const void* w_packed, const float* bias, int cout, int k,
                                           int stride, float slope, float* out, void* stream) {
    return rf_conv3d_valid_leaky_split_ex(x, 0, n, cin, s, w_packed, bias, cout, k, stride, slope, out, 0, stream);
}
