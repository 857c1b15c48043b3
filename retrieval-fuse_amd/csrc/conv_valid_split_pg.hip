// rf_conv3d_valid_leaky_split_pg: the split-operand valid conv of conv_valid_split.hip for the layers the patch encoders evaluate ON THE GRID
// (fully convolutional, model/retrieval.py forward_grid of this package; reference model/retrieval.py:217-243 PCPatch48, layer 12 -> 24 k3 @140^3 of
// config/surface_reconstruction/ShapeNetV2/refinement_128_064.yaml) as a PERSISTENT kernel.  Same operands (x = h + l / 2^11, three MFMAs per product), same K
// order of (tap, 4-channel group) pieces as k_convv_split, on v_mfma_f32_32x32x16_f16 instead of 16x16x32: equal to it within the last bits of an fp32 sum (tests:
// <= 4e-6 on O(1) outputs, both 2e-6 from float64).  What changes is where the operands come from and what runs beside what (every step measured:
// profiles/r06_convv_pg_variants.md):
//
//   k_convv_split, one 256-voxel tile per workgroup: every wave streams the whole weight image (45 KB for 12 -> 24) global -> VGPR per tile -- 29.5 GB per
//       launch through the L2 for a layer whose tensors are 6.1 GB --, per-tile tables and staging latencies, an epilogue through an LDS tile; three
//       workgroups per CU hide some of it: 3.4-3.5 ms, a third of the matrix pipe.
//   here one workgroup per CU (512 threads) walks the tiles as TWO TEAMS of four waves, one wave of each team on every SIMD, half a period apart:
//       while team A runs a tile's k-loop (MFMAs + LDS operand reads), team B runs the previous tile's epilogue and stages its next tile (VALU, global
//       loads / stores, LDS writes), then they swap -- two barriers per tile pair.  The matrix pipe always has a wave in its k-loop; what a first
//       single-team build spent behind it (ablations on the 16 x 140^3 launch: k-loop 1.95 ms, epilogue 0.88, staging 0.38; 3.25 ms in all) runs in
//       its shadow -- the memory part of it: on this part VALU work beside another wave's MFMAs takes issue slots out of that wave's stream, what the teams
//       hide is request / store / LDS-write time.  A team's input tile needs no double buffer: it is re-filled in the team's own non-MFMA phase.
//   The weight image and the tables are copied to LDS once per workgroup; a k-step's operands are requested one k-step ahead, the reads placed BETWEEN
//       the MFMAs (sched_group_barrier: a burst per k-step fills the LDS queue and the MFMA behind it waits for its turn to issue).
//   MFMA roles swapped (weights = A / M = couts, voxels = B / N): an accumulator lane holds 4 consecutive couts of ONE voxel = one 8-byte slot of the
//       split-form output; bias, LeakyReLU, the h / l split and the stores leave from registers (no LDS tile).  Stores are BUFFER stores: a lane
//       whose voxel lies outside the volume (ragged last tiles) or whose channel group is padding gets an out-of-range offset -- no branch.
// Split form in and out only (rf_conv3d_valid_leaky_split_ex's in_split = out_split = 1: the layer sits between two layers that read / write it).
#include "common.h"
#include <type_traits>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr float PG_ACT_SCALE = 1.0f / 16, PG_W_SCALE = 16.0f, PG_LO = 2048.0f;
constexpr int PG_NT = 512, PG_TT = 256, PG_MB = 2;               // threads, threads of a team, 32-voxel m-blocks per wave, output voxels per tile (of a team)
constexpr int PG_SB = 4;                                        // staged items (4 channels of TWO voxels along x: 16 bytes of the h plane, 16 of the l plane) per thread and tile
constexpr size_t PG_LDS_MAX = 160 * 1024;
}   // namespace

struct ConvPGArgs {
    const unsigned char* x;
    const h8* wp;
    const float* bias;
    unsigned char* out;
    int n, cin, s, cout, k, stride, so;
    float slope;
    int tz, ty, tx, ntz, nty, ntx;  // output voxels per tile and dim, tiles per dim
    int zi, yi, xi, npos;           // staged input planes / rows per plane / voxels per row, positions per channel group
    int cg, ksteps;                 // 4-channel groups, k-steps (16 K values = 4 pieces of (tap, channel group) each)
    int items;                      // cg * npos
    int rs, ps, cgs, plane;         // LDS image: bytes between rows / planes / channel groups (ps = 128 mod 256: the two rows of an m-block), bytes of a plane incl. dump slots
    int hdr;                        // ints of the image's table header
    unsigned tiles;                 // n * ntz * nty * ntx
#ifdef RF_PG_DEV
    int ablate;                     // dev build (tools/convv_pg_bench.py, rft_pg_set_ablate): 1 no epilogue, 2 no staging, 4 no k-loop, 8 stamps, 16 no stores
#endif
};

static size_t convv_pg_lds_bytes(const ConvPGArgs& a) {
    const size_t w = (size_t)a.ksteps * 2 * 64 * 16;
    const size_t tables = (size_t)a.hdr * 4 + 32 * 4;
    return w + tables + 2 * 2 * (size_t)a.plane;                 // two teams x (h plane, l plane)
}

// the instantiation built: k = 3, 12 input channels, 17..24 couts, tile 4 x 4 x 16 (a wave = one row index y of the tile, an m-block = that row in two
// adjacent planes); depends on the layer only: the tables of the weight image are made for it
static bool convv_pg_plan(int cin, int s, int cout, int k, int stride, ConvPGArgs& a) {
    // s even: rows of the input start 16-byte aligned and a tile's row has an even number of voxels in the volume (16-byte requests / stores: two voxels)
    if (cin != 12 || cout <= 16 || cout > 24 || (cout & 3) || k != 3 || stride != 1 || s < 64 || s > 254 || (s & 1)) return false;
    const int so = s - k + 1, cg = cin / 4;
    a.cin = cin; a.s = s; a.cout = cout; a.k = k; a.stride = stride; a.so = so; a.cg = cg;
    a.ksteps = (k * k * k * cg + 3) / 4;
    a.tz = 4; a.ty = 4; a.tx = 16;
    a.zi = a.tz - 1 + k; a.yi = a.ty - 1 + k; a.xi = a.tx - 1 + k;
    a.npos = a.zi * a.yi * a.xi; a.items = cg * a.npos;
    a.rs = a.xi * 8;
    a.ps = a.yi * a.rs; a.ps += (128 - a.ps % 256 + 256) % 256;
    a.cgs = a.zi * a.ps;
    a.plane = cg * a.cgs + 32 * 16;
    a.ntz = (so + a.tz - 1) / a.tz; a.nty = (so + a.ty - 1) / a.ty; a.ntx = (so + a.tx - 1) / a.tx;
    a.hdr = rf_round_up(a.ksteps * 4, 4);
    return a.items <= 2 * PG_SB * PG_TT && convv_pg_lds_bytes(a) <= PG_LDS_MAX;
}

extern "C" int rf_conv3d_valid_split_pg_supported(int n, int cin, int s, int cout, int k, int stride) {
    ConvPGArgs a;
    return n > 0 && convv_pg_plan(cin, s, cout, k, stride, a) ? 1 : 0;
}

extern "C" size_t rf_convv_split_pg_packed_bytes(int cout, int cin, int k, int s, int stride) {
    ConvPGArgs a;
    if (!convv_pg_plan(cin, s, cout, k, stride, a)) return 0;
    return (size_t)a.hdr * 4 + (size_t)a.ksteps * 2 * 64 * 16;
}

// image header: [ksteps * 4] byte offset of K slot 4 q + 2 hk + xy = piece (tap, channel group) = tap * cg + group in the h plane of a staged tile
// (slots behind the last piece: zero weights, offset 0)
__global__ void k_convv_pg_header(ConvPGArgs a, int* __restrict__ hdr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.hdr) return;
    const int k = a.k;
    int off = 0;
    if (i < k * k * k * a.cg) {
        const int tap = i / a.cg, cg = i - tap * a.cg;
        off = cg * a.cgs + (tap / (k * k)) * a.ps + ((tap / k) % k) * a.rs + (tap % k) * 8;
    }
    hdr[i] = off;
}

// weight fragments of v_mfma_f32_32x32x16_f16's 32 x 16 operand, [k-step][h | l][lane][8 halves]: lane = 32 hk + cout, K values = slots 4 q + 2 hk, + 1 (4 channels each)
__global__ void k_convv_pg_pack(ConvPGArgs a, const float* __restrict__ w, h8* __restrict__ wp, size_t total) {
    const int cout = a.cout, cin = a.cin, k3 = a.k * a.k * a.k, cg = a.cg;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), part = (int)((i >> 6) & 1), q = (int)(i >> 7);
        const int co = lane & 31, hk = lane >> 5;
        h8 out;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = 4 * q + 2 * hk + (j >> 2);
            double v = 0.0;
            if (p < k3 * cg && co < cout) v = (double)w[((size_t)co * cin + (p % cg) * 4 + (j & 3)) * k3 + p / cg];
            v *= (double)PG_W_SCALE;
            v = v > 65504.0 ? 65504.0 : (v < -65504.0 ? -65504.0 : v);
            const _Float16 h = (_Float16)(float)v;
            out[j] = part == 0 ? h : (_Float16)(float)((v - (double)(float)h) * (double)PG_LO);
        }
        wp[i] = out;
    }
}

extern "C" int rf_convv_split_pg_pack_weight(const float* w_oidhw, int cout, int cin, int k, int s, int stride, void* w_packed, void* stream) {
    RF_REQUIRE(w_oidhw && w_packed, RF_E_INVALID, "rf_convv_split_pg_pack_weight: null pointer");
    ConvPGArgs a;
    RF_REQUIRE(convv_pg_plan(cin, s, cout, k, stride, a), RF_E_UNSUPPORTED,
               "rf_convv_split_pg_pack_weight: layer not taken by the persistent grid form (ask rf_conv3d_valid_split_pg_supported)");
    const size_t total = (size_t)a.ksteps * 2 * 64;
    hipLaunchKernelGGL(k_convv_pg_header, dim3((unsigned)((a.hdr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, reinterpret_cast<int*>(w_packed));
    hipLaunchKernelGGL(k_convv_pg_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, w_oidhw,
                       reinterpret_cast<h8*>(w_packed) + a.hdr / 4, total);
    RF_CHECK_LAUNCH("rf_convv_split_pg_pack_weight");
    return RF_OK;
}

// ------------------------------------------------------------------------------------------------------------------- kernel
namespace {
struct PgTile {             // uniform: origin of a tile
    int nn, z0, y0, x0;
    bool valid;
};
}   // namespace

// dev build (-DRF_PG_DEV): ablations and s_memtime at the phase borders of every workgroup's 9th round (rft_pg_set_ablate bit 3), both teams; tools/convv_pg_bench.py
#ifdef RF_PG_DEV
#define PG_ABL(bit_) (a.ablate & (bit_))
__device__ unsigned long long g_pg_stamps[1024 * 2 * 8];
static int g_pg_ablate = 0;
extern "C" void rft_pg_set_ablate(int bits) { g_pg_ablate = bits; }
extern "C" int rft_pg_read_stamps(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pg_stamps), sizeof(g_pg_stamps)); }
#define PG_STAMP(k_) do { if ((a.ablate & 8) && i == 8 && tw == 0 && lane == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_pg_stamps[(blockIdx.x * 2 + team) * 8 + (k_)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define PG_ABL(bit_) false
#define PG_STAMP(k_) do { } while (0)
#endif

template <int KS, int NRQ, int RS, int PS, int PLANE>
__global__ __launch_bounds__(PG_NT, 1) void k_convv_split_pg(ConvPGArgs a) {
    constexpr int TT = PG_TT, MB = PG_MB, SB = PG_SB;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, ttid = tid & (TT - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wave >> 2, tw = wave & 3;
    const int j = lane & 15, zh = (lane >> 4) & 1, hk = lane >> 5;  // voxel column = lane & 31: x = j of row tw in plane 2 mb + zh; K half of the lane
    const int so = a.so, s = a.s, st = a.stride, items = a.items;
    const size_t ivol = (size_t)s * s * s, ovol = (size_t)so * so * so;

    // LDS: weight fragments | K slot offsets | bias / 16 | per team: h plane, l plane (dump slots behind each)
    const h8* wl = reinterpret_cast<const h8*>(lds);
    constexpr int WBYTES = KS * 2 * 64 * 16;
    int* poff = reinterpret_cast<int*>(lds + WBYTES);
    float* bzt = reinterpret_cast<float*>(poff + ((KS * 4 + 3) & ~3));
    constexpr int plane = PLANE;                                    // (the launcher checks the plan against the instantiation)
    unsigned char* buf = reinterpret_cast<unsigned char*>(bzt + 32) + (size_t)team * 2 * plane;

    // tiles of this workgroup: an XCD (blockIdx & 7) walks a contiguous range of the tile list, its workgroups side by side (shared halos meet in its L2);
    // the workgroup's tiles go to the teams alternately
    const unsigned nxw = gridDim.x >> 3, xk = blockIdx.x & 7u, loc = blockIdx.x >> 3;
    const unsigned per = a.tiles >> 3, rem = a.tiles & 7u;
    const unsigned t_first = xk * per + (xk < rem ? xk : rem), t_count = per + (xk < rem ? 1u : 0u);
    if (loc >= t_count) return;                                     // uniform over the workgroup, in front of every barrier
    const unsigned wg_tiles = (t_count - loc + nxw - 1) / nxw, n_iter = (wg_tiles + 1) >> 1;
    const unsigned tpv = (unsigned)(a.ntz * a.nty * a.ntx);
    // The team's tiles: the workgroup's it = team, team + 2, ...  One decode by division, then mixed-radix steps of 2 * nxw tiles: scalar adds and
    // selects (a uniform 32-bit division is ~30 VALU instructions on gfx9 -- issue slots the other team's MFMAs want).
    auto decode = [&](unsigned lin, unsigned& nn, unsigned& z, unsigned& y, unsigned& x) {
        nn = lin / tpv;
        const unsigned tb = lin - nn * tpv, tzy = tb / (unsigned)a.ntx;
        x = tb - tzy * (unsigned)a.ntx;
        z = tzy / (unsigned)a.nty;
        y = tzy - z * (unsigned)a.nty;
    };
    unsigned w_it = (unsigned)team, w_nn, w_z, w_y, w_x, d_nn, d_z, d_y, d_x;
    decode(t_first + (w_it < wg_tiles ? w_it * nxw + loc : loc), w_nn, w_z, w_y, w_x);
    decode(2 * nxw, d_nn, d_z, d_y, d_x);
    auto walk_tile = [&]() {
        PgTile t;
        t.valid = w_it < wg_tiles;
        t.nn = (int)w_nn; t.z0 = (int)w_z * a.tz; t.y0 = (int)w_y * a.ty; t.x0 = (int)w_x * a.tx;
        return t;
    };
    auto walk_step = [&]() {                                        // behind the last tile: stays on it (requests repeat it, stores are masked)
        if (w_it + 2 < wg_tiles) {
            w_x += d_x;
            unsigned c = w_x >= (unsigned)a.ntx ? 1u : 0u;
            w_x -= c ? (unsigned)a.ntx : 0u;
            w_y += d_y + c;
            c = w_y >= (unsigned)a.nty ? 1u : 0u;
            w_y -= c ? (unsigned)a.nty : 0u;
            w_z += d_z + c;
            c = w_z >= (unsigned)a.ntz ? 1u : 0u;
            w_z -= c ? (unsigned)a.ntz : 0u;
            w_nn += d_nn + c;
        }
        w_it += 2;
    };

    // the slots this thread stages, the same for every tile: item ttid + b * TT = (channel group, input plane, row, PAIR of columns) of the tile's input box
    // -- 16 bytes of the h plane and 16 of the l plane (requests and LDS writes are issue-bound: half as many, twice as wide)
    const int xp = a.xi >> 1, pitems = items >> 1;
    unsigned pk[SB], rel[SB];                                       // packed coordinates; slot offset relative to the tile's first input voxel
    int wofs[SB];                                                   // byte offset of the item's h slots in the team's buffer (items past the tile: dump slots)
#pragma unroll
    for (int b = 0; b < SB; ++b) {
        const int idx = ttid + b * TT;
        const int ic = idx < pitems ? idx : pitems - 1;
        const int cgi = ic / (a.zi * a.yi * xp), r0 = ic - cgi * (a.zi * a.yi * xp);
        const int rz = r0 / (a.yi * xp), r1 = r0 - rz * a.yi * xp;
        const int ry = r1 / xp, ix = 2 * (r1 - ry * xp);
        pk[b] = ((unsigned)cgi << 24) | ((unsigned)rz << 16) | ((unsigned)ry << 8) | (unsigned)ix;
        rel[b] = (unsigned)cgi * 2u * (unsigned)ivol + (unsigned)((rz * s + ry) * s + ix);
        wofs[b] = idx < pitems ? cgi * a.cgs + rz * a.ps + ry * a.rs + ix * 8 : a.cg * a.cgs + (lane & 31) * 16;
    }
    // requests of a tile's slots: planes / rows / columns past the volume (ragged last tiles) are clamped -- only voxels that are not stored see them
    auto request = [&](const PgTile& t, u32x4 (&ph)[SB], u32x4 (&pl)[SB]) {
        const unsigned char* xh = a.x + (size_t)t.nn * a.cg * 2 * ivol * 8;
        const unsigned char* xl = xh + ivol * 8;
        const int zb = t.z0 * st, yb = t.y0 * st, xb = t.x0 * st;
        unsigned off[SB];
        if (zb + a.zi <= s && yb + a.yi <= s && xb + a.xi <= s) {   // uniform; four tiles of five
            const unsigned tb = (unsigned)((zb * s + yb) * s + xb);
#pragma unroll
            for (int b = 0; b < SB; ++b) off[b] = (tb + rel[b]) * 8u;
        } else {
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                int iz = zb + (int)((pk[b] >> 16) & 255u), iy = yb + (int)((pk[b] >> 8) & 255u), ix = xb + (int)(pk[b] & 255u);
                iz = iz < s ? iz : s - 1;
                iy = iy < s ? iy : s - 1;
                ix = ix < s - 1 ? ix : s - 2;
                off[b] = ((pk[b] >> 24) * 2u * (unsigned)ivol + (unsigned)((iz * s + iy) * s + ix)) * 8u;
            }
        }
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            ph[b] = *reinterpret_cast<const u32x4*>(xh + off[b]);
            pl[b] = *reinterpret_cast<const u32x4*>(xl + off[b]);
        }
    };
    auto deposit = [&](const u32x4 (&ph)[SB], const u32x4 (&pl)[SB]) {
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            *reinterpret_cast<u32x4*>(buf + wofs[b]) = ph[b];
            *reinterpret_cast<u32x4*>(buf + plane + wofs[b]) = pl[b];
        }
    };

    // ---- once: weights and tables to LDS, each team's first tile
    {
        const int* hdr = reinterpret_cast<const int*>(a.wp);
        for (int i = tid; i < KS * 4; i += PG_NT) poff[i] = hdr[i];
        const h8* wg = a.wp + a.hdr / 4;
        h8* wd = reinterpret_cast<h8*>(lds);
        for (int i = tid; i < KS * 2 * 64; i += PG_NT) wd[i] = wg[i];
        if (tid < 32) bzt[tid] = (a.bias && tid < a.cout) ? a.bias[tid] * PG_ACT_SCALE : 0.f;
        uint2* zb = reinterpret_cast<uint2*>(bzt + 32);               // the images' padding is never read, the dump slots are never read: tidy all the same
        for (int i = tid; i < 4 * plane / 8; i += PG_NT) zb[i] = make_uint2(0u, 0u);
        __syncthreads();
    }
    // A tile's slots are requested a whole round ahead (under load a request takes 2-3 us to come back -- as long as a phase) and wait in registers
    PgTile cur = walk_tile();
    u32x4 ph[SB], pl[SB];
    request(cur, ph, pl);
    deposit(ph, pl);
    walk_step();
    PgTile nxt = walk_tile();
    request(nxt, ph, pl);
    __syncthreads();
    // input corner of this lane's voxel of m-block 0 in the team's h plane; m-block mb: + 2 mb PS, an immediate -- a k-step costs two address additions.
    // A 32-lane half of a ds_read_b64 reads two rows of 16 voxels (128 bytes each) PS apart: PS = 128 mod 256, no bank is asked twice.
    typedef __attribute__((address_space(3))) h4 lds_h4;
    typedef const volatile lds_h4* lds_vh4p;
    const unsigned bb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)buf + (unsigned)(zh * PS + tw * RS + j * 8);
    const int2* pop = reinterpret_cast<const int2*>(poff) + hk;      // K slot offsets of k-step q for this lane's half: pop[2 q]
    const int eo0 = (zh * so + tw) * so + j;                        // this lane's voxel of m-block 0 in the output volume, relative to the tile's first voxel

    if (team == 1) __builtin_amdgcn_s_barrier();                    // half a period behind team 0 (every wave passes the same number of barriers)

    for (unsigned i = 0; i < n_iter; ++i) {
        // ================= k-loop: v_mfma_f32_32x32x16_f16, weights = A (M = 32 couts), voxels = B (N = 32 voxels); operands of k-step q + 1 requested between
        // the MFMAs of k-step q.  (32 x 32 and not 16 x 16 x 32: a SIMD issues ~4 instructions per 16 cycles for BOTH its waves; half as many MFMAs of twice
        // the length leave the other team's epilogue the slots it needs.)
        PG_STAMP(0);
        f32x16 hi[MB], lo[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { hi[mb][r] = 0.f; lo[mb][r] = 0.f; }
        if (!PG_ABL(4)) {
        h8 bh[2], bl[2], ah[2][MB], al[2][MB];
        int2 po[2];
        auto load_ops = [&](int q, int sl) {
            bh[sl] = wl[(q * 2) * 64 + lane];
            bl[sl] = wl[(q * 2 + 1) * 64 + lane];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const unsigned p0 = bb + (unsigned)po[sl].x + mb * 2 * PS, p1 = bb + (unsigned)po[sl].y + mb * 2 * PS;
                // (volatile: hipcc would merge two reads off one address register into ds_read2_b64 -- half the LDS rate, v_movs to sort the halves into operands)
                const h4 a0 = *(lds_vh4p)(size_t)p0, a1 = *(lds_vh4p)(size_t)p1;
                const h4 c0 = *(lds_vh4p)(size_t)(p0 + plane), c1 = *(lds_vh4p)(size_t)(p1 + plane);
                ah[sl][mb] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                al[sl][mb] = __builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        };
        po[0] = pop[0];
        po[1] = pop[2];
        load_ops(0, 0);
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const int sl = q & 1, sn = sl ^ 1;
            if (q + 1 < KS) load_ops(q + 1, sn);
            if (q + 2 < KS) po[sl] = pop[2 * (q + 2)];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) hi[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[sl], ah[sl][mb], hi[mb], 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) lo[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[sl], ah[sl][mb], lo[mb], 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) lo[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[sl], al[sl][mb], lo[mb], 0, 0, 0);
            if (q + 1 < KS) {                                       // one scheduling region per k-step: the 10-11 reads spread behind the 6 MFMAs
#pragma unroll
                for (int r = 0; r < 3 * MB; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        PG_STAMP(1);
        __builtin_amdgcn_s_barrier();                               // the other team's k-loop starts; this team's buffer is free
        PG_STAMP(2);

        // ================= the team's next tile to LDS (requested a round ago), the one behind it requested, this tile's epilogue from registers
        f32x4 bz_[NRQ];
#pragma unroll
        for (int rq = 0; rq < NRQ; ++rq) bz_[rq] = *reinterpret_cast<const f32x4*>(bzt + (2 * rq + hk) * 4);    // bias / 16 of this lane's channel groups
        if (!PG_ABL(2)) deposit(ph, pl);
        walk_step();
        const PgTile nn2 = walk_tile();
        if (!PG_ABL(2)) request(nn2, ph, pl);                  // (behind the last tile: the last tile again -- harmless, no branch)
        PG_STAMP(3);
        static_assert(PG_ACT_SCALE * PG_W_SCALE == 1.0f, "epilogue assumes the operand scales cancel");
        if (!PG_ABL(1) || hi[0][0] == 123.456f) {
            // accumulator register r of a lane: cout 8 (r / 4) + 4 hk + (r & 3) of its voxel = channel group 2 (r / 4) + hk: rq < NRQ real groups (the couts padded to
            // 32 are whole registers here: never touched)
            int zlim = so - cur.z0, ylim = so - cur.y0, xlim = so - cur.x0;
            zlim = zlim < a.tz ? zlim : a.tz;
            ylim = ylim < a.ty ? ylim : a.ty;
            xlim = xlim < a.tx ? xlim : a.tx;
            if (!cur.valid) zlim = 0;                               // a team without a tile in the last round
            const unsigned obytes = (unsigned)(a.cout >> 2) * 2u * (unsigned)ovol * 8u;
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)cur.nn * obytes, 0, (int)obytes, 0x00020000);
            const unsigned org = (unsigned)((cur.z0 * so + cur.y0) * so + cur.x0) * 8u;
            const unsigned lsoff = (unsigned)ovol * 8u;
            // 8-byte stores, no lane exchange: a 16-byte form (lane pairs trading halves over DPP) was measured -- its 8 extra VALU instructions per store cost more
            // issue slots beside the other team's MFMAs than the halved store count returned.  INTERIOR tiles (four of five; uniform) skip the per-lane bounds.
            auto epilogue = [&](auto interior_tag) {
                constexpr bool INTERIOR = decltype(interior_tag)::value;
                const bool yx_in = INTERIOR || (tw < ylim && j < xlim);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const bool vin = INTERIOR || (yx_in && 2 * mb + zh < zlim);
                    const unsigned vox = org + (unsigned)(eo0 + mb * 2 * so * so) * 8u;
#pragma unroll
                    for (int rq = 0; rq < NRQ; ++rq) {
                        const int grp = 2 * rq + hk;
                        // hi + lo / 2^11 + bias, LeakyReLU, the consumer's 1 / 16, clamp, split -- on register PAIRS (v_pk_*_f32); the 1 / 16 is applied first:
                        // exact, every later step scales with it
                        h4 hh, ll;
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {
                            const f32x2 h2 = {hi[mb][4 * rq + r], hi[mb][4 * rq + r + 1]}, l2 = {lo[mb][4 * rq + r], lo[mb][4 * rq + r + 1]}, b2 = {bz_[rq][r], bz_[rq][r + 1]};
                            const f32x2 t0 = __builtin_elementwise_fma(l2, (f32x2){PG_ACT_SCALE / PG_LO, PG_ACT_SCALE / PG_LO}, h2 * (f32x2){PG_ACT_SCALE, PG_ACT_SCALE}) + b2;
                            const f32x2 ts = t0 * (f32x2){a.slope, a.slope};
                            f32x2 t;
                            t.x = __builtin_amdgcn_fmed3f(fmaxf(t0.x, ts.x), -65504.f, 65504.f);
                            t.y = __builtin_amdgcn_fmed3f(fmaxf(t0.y, ts.y), -65504.f, 65504.f);
                            const f32x2 tl = t * (f32x2){PG_LO, PG_LO};
                            const _Float16 ha = (_Float16)t.x, hb = (_Float16)t.y;
                            hh[r] = ha; hh[r + 1] = hb;
                            ll[r] = (_Float16)fmaf(-PG_LO, (float)ha, tl.x);
                            ll[r + 1] = (_Float16)fmaf(-PG_LO, (float)hb, tl.y);
                        }
                        unsigned vo = (unsigned)grp * 2u * lsoff + vox;
                        if (!INTERIOR) vo = (vin && grp * 4 < a.cout) ? vo : 0xfffffff0u;
                        if (PG_ABL(16)) vo = hh[0] == (_Float16)123.0f ? vo : 0xfffffff0u;
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hh), ro, (int)vo, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ll), ro, (int)vo, (int)lsoff, 0);
                    }
                }
            };
            if (cur.valid && zlim == a.tz && ylim == a.ty && xlim == a.tx && (a.cout >> 2) >= 2 * NRQ) epilogue(std::true_type{});
            else epilogue(std::false_type{});
        }
        PG_STAMP(4);
        cur = nxt;
        nxt = nn2;
        PG_STAMP(5);
        __builtin_amdgcn_s_barrier();                               // the team's next tile is in LDS; the other team's k-loop is over
        PG_STAMP(6);
    }
    if (team == 0) __builtin_amdgcn_s_barrier();
}

// x, out in split form ([n][c/4][h | l][s^3] 8-byte slots); w_packed from rf_convv_split_pg_pack_weight for the same (cout, cin, k, s, stride)
extern "C" int rf_conv3d_valid_leaky_split_pg(const void* x, int n, int cin, int s, const void* w_packed, const float* bias, int cout, int k, int stride,
                                              float slope, void* out, void* stream) {
    RF_REQUIRE(x && w_packed && out && n > 0, RF_E_INVALID, "rf_conv3d_valid_leaky_split_pg: bad arguments");
    RF_REQUIRE(slope >= 0.f && slope <= 1.f, RF_E_INVALID, "rf_conv3d_valid_leaky_split_pg: LeakyReLU slope %g outside [0, 1] (the epilogue takes max(t, slope * t))", (double)slope);
    ConvPGArgs a;
    RF_REQUIRE(convv_pg_plan(cin, s, cout, k, stride, a), RF_E_UNSUPPORTED,
               "rf_conv3d_valid_leaky_split_pg: layer not taken by the persistent grid form (ask rf_conv3d_valid_split_pg_supported)");
    a.n = n; a.x = reinterpret_cast<const unsigned char*>(x); a.wp = reinterpret_cast<const h8*>(w_packed); a.bias = bias; a.out = reinterpret_cast<unsigned char*>(out);
    a.slope = slope;
#ifdef RF_PG_DEV
    a.ablate = g_pg_ablate;
#endif
    const size_t tiles64 = (size_t)a.ntz * a.nty * a.ntx * n;
    RF_REQUIRE(tiles64 < (1ull << 31), RF_E_INVALID, "rf_conv3d_valid_leaky_split_pg: too many tiles (%zu)", tiles64);
    RF_REQUIRE((size_t)cin / 4 * 2 * s * s * s * 8 < (1ull << 31) && (size_t)cout / 4 * 2 * a.so * a.so * a.so * 8 < (1ull << 31), RF_E_INVALID,
               "rf_conv3d_valid_leaky_split_pg: a sample does not fit 32-bit offsets");
    a.tiles = (unsigned)tiles64;
    unsigned grid = (unsigned)(rf_resident_wgs() / 2) & ~7u;        // one workgroup per CU, whole XCD groups
    if (grid < 8) grid = 8;
    while ((size_t)grid * 2 > tiles64 && grid > 8) grid -= 8;
    const size_t lds = convv_pg_lds_bytes(a);
    static RfLdsOptIn opt_in;
    RF_REQUIRE(a.ksteps == 21 && a.rs == 144 && a.ps == 896 && a.plane == 16640 && a.cout > 16 && a.cout <= 24, RF_E_UNSUPPORTED,
               "rf_conv3d_valid_leaky_split_pg: plan differs from the instantiation built");
    if (int rc = opt_in.ensure(reinterpret_cast<const void*>(k_convv_split_pg<21, 3, 144, 896, 16640>), (int)PG_LDS_MAX, "rf_conv3d_valid_leaky_split_pg")) return rc;
    hipLaunchKernelGGL((k_convv_split_pg<21, 3, 144, 896, 16640>), dim3(grid), dim3(PG_NT), lds, (hipStream_t)stream, a);
    RF_CHECK_LAUNCH("rf_conv3d_valid_leaky_split_pg");
    return RF_OK;
}
