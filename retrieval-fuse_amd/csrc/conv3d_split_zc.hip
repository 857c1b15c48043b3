// k_conv3_split_zc: SingleConv 'gcr' (reference model/unet.py:19-76) of a level-0 DoubleConv's SECOND conv -- 8 -> 16 channels on 16^3 samples, input
// pre-split by its producer (rf_conv3d_cin1_presplit) -- as a PERSISTENT workgroup that walks whole samples, box by box, with the operand
// reads and the weight traffic of the box kernel (conv3d_split.hip) cut down.  Same arithmetic contract: exact f16 x f16 products of split operands,
// fp32 accumulation in separate hi / lo accumulators, combined once.
//
// What is different from k_conv3_split<1, 6, true, false, true> (round 3: 0.75 ms alone, 1.06-1.18 ms in the step, MFMA pipe 26-40 % busy):
//  * Z-COLUMNS with operand reuse.  A wave owns 4 z-planes of a 2 (y) x 8 (x) footprint (m-block = one plane) and the 27 taps are ordered so that a
//    k-step is four taps of ONE dz: k-step (dz, q) = taps (dz, j = 4q + g), j = (dy + 1) * 3 + dx + 1 < 8, lane group g; the ninth (dy, dx) = (1, 1) of the
//    three dz make k-step 6.  The A operand "image plane P, half q" then serves m-block P (as dz = -1), P - 1 (dz = 0) and P - 2 (dz = +1): 16 operand
//    reads per wave and box instead of 28, each feeding up to nine MFMAs instead of three (the 16-cout box kernel asked the LDS for an A fragment per
//    48 matrix-pipe cycles of every SIMD).  The weight image is the box kernel's; the workgroup permutes it into this tap order while copying it to LDS.
//  * PERSISTENT: a workgroup keeps the layer's weight fragments (14 KB) in LDS for its whole life -- no weight load in the box loop, so no s_waitcnt vmcnt
//    between MFMAs that would also wait for the next box's voxels (vmcnt retires in order) -- and stages box i + 1 (two halo voxels per thread, copies of
//    16-byte slots) into the other image buffer while box i multiplies.
//  * x-adjacent box PAIRS share one epilogue: the pooled rows of both boxes go through an LDS tile and leave as 128-byte runs (the box kernel wrote
//    16-byte fragments: 4.2x write amplification on the pooled-only layer); GroupNorm statistics are accumulated per wave over the eight
//    boxes of a sample and reduced once per sample (stats_tiles = 1).
#include "common.h"
#include "conv_box.h"
#include "conv_split_common.h"
#include <type_traits>

namespace {
constexpr int ZC_W_BYTES = 7 * 2 * 64 * 16;                       // 14,336: [k-step][h | l][lane][8 halves]
constexpr int ZC_SCRATCH = ZC_W_BYTES + 2 * CS_BUF;               // 512 bytes behind the images: per-cout sums and GroupNorm triples of the pre-split hand-over
constexpr int ZC_LDS_BYTES = ZC_SCRATCH + 512;                    // 81,408: two workgroups per CU
constexpr int ZC_TILE_STRIDE = 132;                               // floats per cout row of the pooled tile (128 + 4: conflict-free float2 writes)
constexpr int ZC_TILE_BYTES = 16 * ZC_TILE_STRIDE * 4;            // 8,448
constexpr int ZC_RED_BYTES = 8 * 16 * 16;                         // [wave][cout] double2
static_assert(ZC_TILE_BYTES + 2 * ZC_RED_BYTES <= CS_BUF, "epilogue scratch lives in a dead image buffer");

__device__ __forceinline__ void lds_barrier() {                   // LDS-only: global loads of the next box stay in flight across it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct ZcStage { h8 ph[2], pl[2]; bool in[2]; };
}   // namespace

template <bool FULL>                                                // FULL: the full-resolution output (and its statistics) too; false: pooled only
__global__ __launch_bounds__(512, 4) void k_conv3_split_zc(ConvArgs a, SplitPreOut po, int samples_per_wg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int E = 16, VOL = E * E * E, PE = 8, PVOL = PE * PE * PE;
    const int n_first = blockIdx.x * samples_per_wg;
    const int n_last = min(a.n, n_first + samples_per_wg);
    if (n_first >= n_last) return;

    // ---- weights: the box kernel's image [k-step s][n-block][h | l][lane = 16 g + cout][8 halves] with tap 4 s + g, copied to LDS in THIS kernel's order
    {
        const int nbt = a.cout16 >> 4;
        const h8* __restrict__ wsrc = reinterpret_cast<const h8*>(a.wp) + (size_t)blockIdx.y * 128;
        for (int f = tid; f < 7 * 128; f += 512) {
            const int k = f >> 7, piece = (f >> 6) & 1, g = (f >> 4) & 3, co = f & 15;
            int tap;
            if (k < 6) tap = (k >> 1) * 9 + 4 * (k & 1) + g;
            else tap = g < 3 ? g * 9 + 8 : 27;                      // tap 27: the box image's zero dummy
            reinterpret_cast<h8*>(lds)[f] = wsrc[((size_t)(tap >> 2) * nbt * 2 + piece) * 64 + (tap & 3) * 16 + co];
        }
    }

    // ---- staging: thread t owns halo voxels t and t + 512 (the second only for t < 488) of every box
    int hpos[2], vslot[2];                                          // halo position (hz, hy, hx) packed 8 bits each; byte offset of the slot in a plane
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int v = tid + r * 512;
        const int hx = v % 10, hy = (v / 10) % 10, hz = v / 100;
        hpos[r] = hz << 16 | hy << 8 | hx;
        vslot[r] = (hz * CS_SZ + hy * CS_SY + hx) * 16;
    }
    const unsigned char* __restrict__ spre = reinterpret_cast<const unsigned char*>(a.src0);
    auto stage_load = [&](ZcStage& st, int box) {                   // box = sample * 8 + (bz, by, bx)
        const int n0 = box >> 3, z0 = ((box >> 2) & 1) * 8 - 1, y0 = ((box >> 1) & 1) * 8 - 1, x0 = (box & 1) * 8 - 1;
        const unsigned char* p = spre + (size_t)n0 * (2 * VOL * 16);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int z = z0 + (hpos[r] >> 16), y = y0 + ((hpos[r] >> 8) & 255), x = x0 + (hpos[r] & 255);
            st.in[r] = (unsigned)z < (unsigned)E && (unsigned)y < (unsigned)E && (unsigned)x < (unsigned)E && (r == 0 || tid < CS_VOX - 512);
            const int off = st.in[r] ? ((z * E + y) * E + x) * 16 : 0;
            st.ph[r] = *reinterpret_cast<const h8*>(p + off);
            st.pl[r] = *reinterpret_cast<const h8*>(p + VOL * 16 + off);
        }
    };
    auto stage_store = [&](const ZcStage& st, unsigned char* img) {  // zeros outside the volume (the padding of the NORMALISED tensor)
        const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (r == 0 || tid < CS_VOX - 512) {
                *reinterpret_cast<h8*>(img + vslot[r]) = st.in[r] ? st.ph[r] : zero;
                *reinterpret_cast<h8*>(img + vslot[r] + CS_PLANE) = st.in[r] ? st.pl[r] : zero;
            }
    };

    // ---- operand addressing.  wave = (zh, yq): planes z = 4 zh + mb, rows y = 2 yq + (ri >> 3), x = ri & 7; halo slot (z + 1, y + 1, x + 1)
    const int g = lane >> 4, ri = lane & 15, zh = wave >> 2, yq = wave & 3;
    int aq[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = 4 * q + g, dy = j / 3 - 1, dx = j % 3 - 1;
        aq[q] = ((4 * zh) * CS_SZ + (2 * yq + (ri >> 3) + dy + 1) * CS_SY + (ri & 7) + dx + 1) * 16;      // + P * CS_SZ * 16: image plane 4 zh + P
    }
    const int a6 = ((4 * zh + (g < 3 ? g : 2)) * CS_SZ + (2 * yq + (ri >> 3) + 2) * CS_SY + (ri & 7) + 2) * 16;   // + mb * CS_SZ * 16; lane group 3: zero weights
    const unsigned char* const wl = lds + lane * 16;
    constexpr int PSTEP = CS_SZ * 16;

    auto mf = [](const h8& x, const h8& y, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c, 0, 0, 0); };
    // all MFMAs of one box for this wave: acc = hi + lo / 2^11 of the four planes
    auto compute = [&](const unsigned char* img, f32x4 (&acc)[4]) {
        f32x4 hi[4], lo[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) { hi[m] = (f32x4){0.f, 0.f, 0.f, 0.f}; lo[m] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        h8 fh[2], fl[2];
        h8 wh[3], wlo[3];
        {
            const unsigned char* ap = img + aq[0];
            fh[0] = *reinterpret_cast<const h8*>(ap);
            fl[0] = *reinterpret_cast<const h8*>(ap + CS_PLANE);
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            wh[d] = *reinterpret_cast<const h8*>(wl + (2 * d) * 2048);
            wlo[d] = *reinterpret_cast<const h8*>(wl + (2 * d) * 2048 + 1024);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned char* ap = img + aq[q];
#pragma unroll
            for (int P = 0; P < 6; ++P) {
                const int cur = (q * 6 + P) & 1, nxt = cur ^ 1;
                // next operand: plane P + 1 of this half, the first plane of the other half, or the first operand of k-step 6
                const unsigned char* np = P < 5 ? ap + (P + 1) * PSTEP : (q == 0 ? img + aq[1] : img + a6);
                fh[nxt] = *reinterpret_cast<const h8*>(np);
                fl[nxt] = *reinterpret_cast<const h8*>(np + CS_PLANE);
                __builtin_amdgcn_sched_barrier(0);
                // plane P is tap dz = -1 of m-block P, dz = 0 of m-block P - 1, dz = +1 of m-block P - 2 (weights d = dz + 1)
#pragma unroll
                for (int d = 0; d < 3; ++d) { const int m = P - d; if (m >= 0 && m < 4) hi[m] = mf(fh[cur], wh[d], hi[m]); }
#pragma unroll
                for (int d = 0; d < 3; ++d) { const int m = P - d; if (m >= 0 && m < 4) lo[m] = mf(fh[cur], wlo[d], lo[m]); }
#pragma unroll
                for (int d = 0; d < 3; ++d) { const int m = P - d; if (m >= 0 && m < 4) lo[m] = mf(fl[cur], wh[d], lo[m]); }
                __builtin_amdgcn_sched_barrier(0);
                // weights of the other half / of k-step 6 into the registers whose k-step has had its last use
                if (q == 0 && P >= 3) {
                    const int d = P - 3;
                    wh[d] = *reinterpret_cast<const h8*>(wl + (2 * d + 1) * 2048);
                    wlo[d] = *reinterpret_cast<const h8*>(wl + (2 * d + 1) * 2048 + 1024);
                }
                if (q == 1 && P == 3) {
                    wh[0] = *reinterpret_cast<const h8*>(wl + 6 * 2048);
                    wlo[0] = *reinterpret_cast<const h8*>(wl + 6 * 2048 + 1024);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {                                 // k-step 6: (dy, dx) = (1, 1) of the three dz of m-block m
            const int cur = m & 1, nxt = cur ^ 1;
            if (m < 3) {
                fh[nxt] = *reinterpret_cast<const h8*>(img + a6 + (m + 1) * PSTEP);
                fl[nxt] = *reinterpret_cast<const h8*>(img + a6 + (m + 1) * PSTEP + CS_PLANE);
            }
            __builtin_amdgcn_sched_barrier(0);
            hi[m] = mf(fh[cur], wh[0], hi[m]);
            lo[m] = mf(fh[cur], wlo[0], lo[m]);
            lo[m] = mf(fl[cur], wh[0], lo[m]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][r] = fmaf(lo[m][r], 1.0f / CS_LO, hi[m][r]);
    };

    // ---- outputs of one box from the accumulators.  D tile: col = lane & 15 = cout, rows 4 (lane >> 4) + r = voxel i: x = i & 7, y = 2 yq + (i >> 3)
    const int col = lane & 15, kq = lane >> 4;
    const bool want_full = FULL, want_pool = a.pool_mode != 0;
    double fsm = 0.0, fsq = 0.0, psm = 0.0, psq = 0.0;              // this lane's share of the sample's statistics (full output / pooled output)
    auto box_outputs = [&](const f32x4 (&acc)[4], int box, float2 (&pooled)[2]) {
        const int n0 = box >> 3, z0 = ((box >> 2) & 1) * 8, y0 = ((box >> 1) & 1) * 8, x0 = (box & 1) * 8;
        if (want_full) {
            if (col < a.cout) {
                float* o = a.out + ((size_t)n0 * a.cout + col) * VOL + (size_t)(y0 + 2 * yq + (kq >> 1)) * E + x0 + 4 * (kq & 1);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const f32x4 v = acc[m];
                    *reinterpret_cast<float4*>(o + (size_t)(z0 + 4 * zh + m) * E * E) =
                        make_float4(fmaxf(v[0], a.floor), fmaxf(v[1], a.floor), fmaxf(v[2], a.floor), fmaxf(v[3], a.floor));
                }
            }
            if (a.stats) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const double v = (double)fmaxf(acc[m][r], 0.f); fsm += v; fsq += v * v; }
            }
        }
        if (want_pool) {                                              // z pair = planes (2 zp, 2 zp + 1), x pairs in r, y pair in lane ^ 32
#pragma unroll
            for (int zp = 0; zp < 2; ++zp) {
                const f32x4 u = acc[2 * zp], v = acc[2 * zp + 1];
                float p0 = fmaxf(fmaxf(u[0], u[1]), fmaxf(v[0], v[1]));
                float p1 = fmaxf(fmaxf(u[2], u[3]), fmaxf(v[2], v[3]));
                p0 = fmaxf(p0, __shfl_xor(p0, 32, 64));
                p1 = fmaxf(p1, __shfl_xor(p1, 32, 64));
                p0 = fmaxf(p0, 0.f);                                  // max and ReLU commute
                p1 = fmaxf(p1, 0.f);
                pooled[zp] = make_float2(p0, p1);
                if (kq < 2) { psm += (double)p0 + (double)p1; psq += (double)p0 * (double)p0 + (double)p1 * (double)p1; }
            }
        }
    };

    // ---- the box loop: pairs of x-adjacent boxes; the image buffers swap roles every pair
    unsigned char* bufA = lds + ZC_W_BYTES;                         // holds box A of the current pair
    unsigned char* bufB = lds + ZC_W_BYTES + CS_BUF;
    ZcStage st;
    stage_load(st, n_first * 8);
    stage_store(st, bufA);
    lds_barrier();                                                  // weights and the first image are in place
    const int pair_end = n_last * 4;
    for (int pair = n_first * 4; pair < pair_end; ++pair) {
        const int boxA = 2 * pair, boxB = boxA + 1;
        f32x4 acc[4];
        float2 poolA[2], poolB[2];
        stage_load(st, boxB);
        __builtin_amdgcn_sched_barrier(0);
        compute(bufA, acc);
        box_outputs(acc, boxA, poolA);
        stage_store(st, bufB);
        lds_barrier();                                              // 1: box B's image complete, box A's image dead
        const int next = pair + 1 < pair_end ? boxB + 1 : boxB;     // (the last pair re-loads its own box: no branch around the loads)
        stage_load(st, next);
        __builtin_amdgcn_sched_barrier(0);
        compute(bufB, acc);
        box_outputs(acc, boxB, poolB);
        const bool sample_end = (pair & 3) == 3;
        float* tile = reinterpret_cast<float*>(bufA);
        double2* red = reinterpret_cast<double2*>(bufA + ZC_TILE_BYTES);
        if (want_pool && kq < 2) {
#pragma unroll
            for (int zp = 0; zp < 2; ++zp) {
                float* t = tile + col * ZC_TILE_STRIDE + ((2 * zh + zp) * 4 + yq) * 8 + 2 * kq;
                *reinterpret_cast<float2*>(t) = poolA[zp];
                *reinterpret_cast<float2*>(t + 4) = poolB[zp];
            }
        }
        if (sample_end) {
            if (a.pool_stats || po.out) {                              // (the hand-over needs the pooled sample's sums whether or not the caller wants them)
                const double s1 = psm + __shfl_xor(psm, 16, 64), s2 = psq + __shfl_xor(psq, 16, 64);      // the two x halves (lane groups 0 and 1)
                if (lane < 16) red[wave * 16 + lane] = make_double2(s1, s2);
            }
            if (FULL && a.stats) {
                double s1 = fsm + __shfl_xor(fsm, 16, 64), s2 = fsq + __shfl_xor(fsq, 16, 64);
                s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                if (lane < 16) red[128 + wave * 16 + lane] = make_double2(s1, s2);
            }
            fsm = fsq = psm = psq = 0.0;
        }
        lds_barrier();                                              // 2: tile (and the sample's partial sums) complete; box B's image dead
        // everything below is addressed from an opaque copy of the thread index: the lane-dependent parts of these addresses do not depend on the pair, and
        // hipcc otherwise forms them ahead of the loop and keeps them in scratch across the MFMAs (21 spilled registers, reloaded here one by one)
        int tv = tid;
        asm volatile("" : "+v"(tv));
        if (want_pool) {
            const int co = tv >> 5, pz = (tv >> 3) & 3, w4 = tv & 7;
            if (co < a.cout) {
                const int n0 = boxA >> 3, z0 = ((boxA >> 2) & 1) * 4, y0 = ((boxA >> 1) & 1) * 4;
                const float4 v = *reinterpret_cast<const float4*>(tile + co * ZC_TILE_STRIDE + pz * 32 + w4 * 4);
                // hand-over mode (po.out): pool_out is a per-workgroup scratch slot -- the values are read back at the sample's end and nobody else wants them,
                // so 32 KB per workgroup stay in L2 instead of a [n][cout][8^3] tensor streaming to HBM and back (0.27 GB each way per launch)
                const size_t slot = po.out ? (size_t)blockIdx.x : (size_t)n0;
                *reinterpret_cast<float4*>(a.pool_out + (slot * a.cout + co) * PVOL + (size_t)(z0 + pz) * 64 + y0 * 8 + w4 * 4) = v;
            }
        }
        if (sample_end && tv < 32) {
            const int which = tv >> 4, co = tv & 15;
            double2* dst = which ? a.stats : a.pool_stats;
            if ((FULL || !which) && (dst || (!which && po.out)) && co < a.cout) {
                double sm = 0.0, sq = 0.0;
#pragma unroll
                for (int w = 0; w < 8; ++w) { const double2 v = red[which * 128 + w * 16 + co]; sm += v.x; sq += v.y; }
                if (dst) dst[(size_t)(boxA >> 3) * a.cout + co] = make_double2(sm, sq);
                if (!which && po.out) reinterpret_cast<double2*>(lds + ZC_SCRATCH)[co] = make_double2(sm, sq);
            }
        }
        if (sample_end && po.out) {
            // the POOLED sample handed to the next level's first conv pre-split: the workgroup has walked the whole sample, so it has the
            // statistics that layer's GroupNorm needs; the pooled values (32 KB, just written, L2-resident) are read back, normalised, split and stored as
            // [2 channel groups][h | l][8^3][8 halves] -- the consumer stages copies and rf_gn_from_stats is not launched
            double2* chst = reinterpret_cast<double2*>(lds + ZC_SCRATCH);
            float4* trip = reinterpret_cast<float4*>(lds + ZC_SCRATCH + 256);
            // this workgroup's pooled stores have reached L2 (stores retire through vmcnt); the read-back below goes to L2 past the L1 (device-scope relaxed
            // atomic loads).  NOT a device-scope release / acquire fence pair: that writes back and invalidates whole caches -- measured 0.57 -> 3.4 ms per launch
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            if (tv < a.cout) {                                        // as rf_gn_from_stats: group sums in channel order, float64
                const int cpg = a.cout / po.groups, c0 = (tv / cpg) * cpg;
                double sm = 0.0, sq = 0.0;
                for (int c = c0; c < c0 + cpg; ++c) { sm += chst[c].x; sq += chst[c].y; }
                const double count = (double)cpg * PVOL, mean = sm / count;
                double var = sq / count - mean * mean;
                if (var < 0.0) var = 0.0;
                trip[tv] = gn_affine(mean, 1.0 / sqrt(var + (double)po.eps), po.gamma[tv], po.beta[tv]);
            }
            lds_barrier();
            const int n0 = boxA >> 3;
            const float* pv = a.pool_out + (size_t)blockIdx.x * a.cout * PVOL + (unsigned)tv;       // this workgroup's scratch slot (see the pooled stores)
            h8* __restrict__ o = po.out + (size_t)n0 * (a.cout >> 3) * 2 * PVOL + (unsigned)tv;
            for (int sg = 0; sg < (a.cout >> 3); ++sg) {
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 t4 = trip[sg * 8 + j];
                    y[j] = fmaf(__hip_atomic_load(pv + (sg * 8 + j) * PVOL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - t4.x, t4.y, t4.z);
                }
                h8 h, l;
                cs_split8(y, h, l);
                o[(size_t)sg * 2 * PVOL] = h;
                o[(size_t)sg * 2 * PVOL + PVOL] = l;
            }
        }
        stage_store(st, bufB);                                      // the next pair's box A into the buffer box B has left
        lds_barrier();                                              // 3: ... complete; the tile has been read
        unsigned char* t = bufA; bufA = bufB; bufB = t;
    }
}

// ------------------------------------------------------------------------------------------------- multi-chunk layers, any edge
// k_conv3_split_zcm: the same persistent z-column form for layers of SEVERAL 8-channel chunks with up to 16 couts (56 -> 16 @8^3 of the retrieval
// backbone, 16 -> 16 @64^3 of the final decoder): the work items of a workgroup are (box, chunk) pairs, the accumulators live across the chunks of a box.
// A chunk's weight fragments (14 KB) sit in LDS in two halves -- R0 = the k-steps of half q = 0, R1 = those of q = 1 and k-step 6 -- and beside the two
// image buffers there is room for ONE chunk of them (two workgroups per CU: 81,920 bytes each), so a half is replaced as soon as every wave has left it:
//     request R1 of THIS chunk | request the next item's voxels | pass 0 (R0) | store R1 | MID BARRIER | request R0 of the NEXT chunk |
//     pass 1 and k-step 6 (R1) | store the voxels, store R0 | END BARRIER
// The weight request always precedes the voxel requests it has to overtake (vmcnt retires in order): waiting for weights never waits for voxels.
// PRE: pre-split input (staging = copies); otherwise the fp32 input is normalised (GroupNorm triples) and split while staged.
// EPI 0: full-resolution output (float4 rows) and / or the fused MaxPool3d(2), optional statistics per box; EPI 1: the final decoder's pointwise head + tanh
// (SplitPreOut.pw_*); EPI 2: whole 8^3 samples handed to the next layer pre-split (SplitPreOut.out / gamma / beta / groups / eps).
namespace {
constexpr int ZM_R0 = 0, ZM_R1 = 3 * 2048, ZM_IMG = 7 * 2048;    // LDS: R0 (k-steps 0, 2, 4), R1 (1, 3, 5, 6), two image buffers
constexpr int ZM_SCRATCH = ZM_IMG + 2 * CS_BUF;                   // 1 KB behind the images: per-cout sums and GroupNorm triples of the pre-split epilogue
constexpr int ZM_LDS_BYTES = ZM_SCRATCH + 1024;                   // 81,920: exactly half a CU's LDS
constexpr int ZM_TILE_STRIDE = 517;                               // pointwise epilogue: [16 couts][8^3] floats, odd stride (conflict-free scalar writes)
static_assert(16 * ZM_TILE_STRIDE * 4 <= CS_BUF, "the epilogue tile lives in a dead image buffer");
}   // namespace

template <bool PRE, int EPI, bool CH8 = false>                   // CH8: the fp32 input is channel-interleaved, [n][cin / 8][voxel][8] (two 16-byte loads per voxel)
__global__ __launch_bounds__(512, 4) void k_conv3_split_zcm(ConvArgs a, SplitPreOut po, int boxes_per_wg, int total_boxes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int edge = a.edge, cin = a.c0, nC = cin >> 3;
    const int tsh = __builtin_ctz(edge >> 3);                       // boxes per axis = 1 << tsh
    const size_t vol = (size_t)edge * edge * edge;
    const int b_first = blockIdx.x * boxes_per_wg;
    const int b_last = min(total_boxes, b_first + boxes_per_wg);
    if (b_first >= b_last) return;
    auto box_origin = [&](int box, int& n0, int& z0, int& y0, int& x0) {
        const int m = (1 << tsh) - 1;
        x0 = (box & m) << 3; y0 = ((box >> tsh) & m) << 3; z0 = ((box >> (2 * tsh)) & m) << 3; n0 = box >> (3 * tsh);
    };

    // ---- weights: per chunk the box kernel's image [k-step s][n-block][h | l][lane][8 halves] (tap 4 s + g), permuted into this kernel's k-steps while
    // copied: thread t moves fragment slot t of R0 (t < 384) and slot t of R1
    const int nbt = a.cout16 >> 4;
    const h8* __restrict__ wsrc = reinterpret_cast<const h8*>(a.wp) + (size_t)blockIdx.y * 128;
    const int wstride = 7 * nbt * 128;                              // h8 per chunk
    int wsrc0, wsrc1;
    {
        const int j = tid >> 7, piece = (tid >> 6) & 1, g4 = (tid >> 4) & 3, co = tid & 15;
        const int tap0 = (j < 3 ? j : 2) * 9 + g4;                  // k-step 2 j: dz = j - 1, half 0
        const int tap1 = j < 3 ? j * 9 + 4 + g4 : (g4 < 3 ? g4 * 9 + 8 : 27);      // k-steps 1, 3, 5 and 6 (tap 27: the image's zero dummy)
        wsrc0 = ((tap0 >> 2) * nbt * 2 + piece) * 64 + (tap0 & 3) * 16 + co;
        wsrc1 = ((tap1 >> 2) * nbt * 2 + piece) * 64 + (tap1 & 3) * 16 + co;
    }

    // ---- staging: thread t owns halo voxels t and t + 512 (the second only for t < 488)
    int hpos[2], vslot[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int v = tid + r * 512;
        const int hx = v % 10, hy = (v / 10) % 10, hz = v / 100;
        hpos[r] = hz << 16 | hy << 8 | hx;
        vslot[r] = (hz * CS_SZ + hy * CS_SY + hx) * 16;
    }
    struct Stage { h8 ph[PRE ? 2 : 1], pl[PRE ? 2 : 1]; float x[PRE ? 1 : 2][PRE ? 1 : 8]; bool in[2]; };
    auto stage_load = [&](Stage& st, int box, int ca) {
        int n0, z0, y0, x0;
        box_origin(box, n0, z0, y0, x0);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int z = z0 - 1 + (hpos[r] >> 16), y = y0 - 1 + ((hpos[r] >> 8) & 255), x = x0 - 1 + (hpos[r] & 255);
            st.in[r] = (unsigned)z < (unsigned)edge && (unsigned)y < (unsigned)edge && (unsigned)x < (unsigned)edge && (r == 0 || tid < CS_VOX - 512);
            // uniform 64-bit base per (sample, chunk) + 32-bit lane offsets: address registers the compiler does not have to carry (or spill) as pairs
            unsigned off = st.in[r] ? (unsigned)((z * edge + y) * edge + x) : 0u;
            if constexpr (PRE) {
                if (a.src_pm) off = st.in[r] ? (unsigned)((((z & 1) * 4 + (y & 1) * 2 + (x & 1)) << 6) + ((z >> 1) << 4) + ((y >> 1) << 2) + (x >> 1)) : 0u;      // edge 8 only
                const unsigned char* p = reinterpret_cast<const unsigned char*>(a.src0) + ((size_t)n0 * nC + ca) * 2 * vol * 16;
                st.ph[r] = *reinterpret_cast<const h8*>(p + off * 16u);
                st.pl[r] = *reinterpret_cast<const h8*>(p + ((unsigned)vol + off) * 16u);
            } else if constexpr (CH8) {
                const float4* p = reinterpret_cast<const float4*>(a.src0 + (((size_t)n0 * nC + ca) * vol << 3)) + off * 2u;
                const float4 u = p[0], v = p[1];
                st.x[r][0] = u.x; st.x[r][1] = u.y; st.x[r][2] = u.z; st.x[r][3] = u.w;
                st.x[r][4] = v.x; st.x[r][5] = v.y; st.x[r][6] = v.z; st.x[r][7] = v.w;
            } else {
                const float* p = a.src0 + ((size_t)n0 * cin + ca * 8) * vol;
#pragma unroll
                for (int j = 0; j < 8; ++j) st.x[r][j] = p[(unsigned)j * (unsigned)vol + off];
            }
        }
    };
    auto stage_store = [&](const Stage& st, int box, int ca, unsigned char* img) {      // zeros outside the volume (the padding of the NORMALISED tensor)
        const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        float4 af[PRE ? 1 : 8];
        if constexpr (!PRE) {
            const int n0 = box >> (3 * tsh);
#pragma unroll
            for (int j = 0; j < 8; ++j) af[j] = a.affine[(size_t)n0 * cin + ca * 8 + j];       // uniform: scalar loads
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            h8 h, l;
            if constexpr (PRE) {
                h = st.in[r] ? st.ph[r] : zero;
                l = st.in[r] ? st.pl[r] : zero;
            } else {
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = st.in[r] ? fmaf(st.x[r][j] - af[j].x, af[j].y, af[j].z) : 0.f;
                cs_split8(y, h, l);
            }
            if (r == 0 || tid < CS_VOX - 512) {
                *reinterpret_cast<h8*>(img + vslot[r]) = h;
                *reinterpret_cast<h8*>(img + vslot[r] + CS_PLANE) = l;
            }
        }
    };

    // ---- operand addressing (as k_conv3_split_zc)
    const int g = lane >> 4, ri = lane & 15, zh = wave >> 2, yq = wave & 3;
    int aq[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = 4 * q + g, dy = j / 3 - 1, dx = j % 3 - 1;
        aq[q] = ((4 * zh) * CS_SZ + (2 * yq + (ri >> 3) + dy + 1) * CS_SY + (ri & 7) + dx + 1) * 16;
    }
    const int a6 = ((4 * zh + (g < 3 ? g : 2)) * CS_SZ + (2 * yq + (ri >> 3) + 2) * CS_SY + (ri & 7) + 2) * 16;
    const unsigned char* const wl = lds + lane * 16;
    constexpr int PSTEP = CS_SZ * 16;
    auto mf = [](const h8& x, const h8& y, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c, 0, 0, 0); };

    f32x4 hi[4], lo[4];
    // half q of a chunk: planes P = 0..5 of the wave's column; plane P is tap dz = -1 of m-block P, dz = 0 of P - 1, dz = +1 of P - 2
    auto half_pass = [&](const unsigned char* img, auto qc) {
        constexpr int q = decltype(qc)::value;
        const unsigned char* wr = wl + (q == 0 ? ZM_R0 : ZM_R1);
        h8 wh[3], wlo[3], fh[2], fl[2];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            wh[d] = *reinterpret_cast<const h8*>(wr + d * 2048);
            wlo[d] = *reinterpret_cast<const h8*>(wr + d * 2048 + 1024);
        }
        const unsigned char* ap = img + aq[q];
        fh[0] = *reinterpret_cast<const h8*>(ap);
        fl[0] = *reinterpret_cast<const h8*>(ap + CS_PLANE);
#pragma unroll
        for (int P = 0; P < 6; ++P) {
            const int cur = P & 1, nxt = cur ^ 1;
            if (P < 5) {
                fh[nxt] = *reinterpret_cast<const h8*>(ap + (P + 1) * PSTEP);
                fl[nxt] = *reinterpret_cast<const h8*>(ap + (P + 1) * PSTEP + CS_PLANE);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < 3; ++d) { const int m = P - d; if (m >= 0 && m < 4) hi[m] = mf(fh[cur], wh[d], hi[m]); }
#pragma unroll
            for (int d = 0; d < 3; ++d) { const int m = P - d; if (m >= 0 && m < 4) lo[m] = mf(fh[cur], wlo[d], lo[m]); }
#pragma unroll
            for (int d = 0; d < 3; ++d) { const int m = P - d; if (m >= 0 && m < 4) lo[m] = mf(fl[cur], wh[d], lo[m]); }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto kstep6 = [&](const unsigned char* img) {
        h8 fh[2], fl[2];
        const h8 w6h = *reinterpret_cast<const h8*>(wl + ZM_R1 + 3 * 2048), w6l = *reinterpret_cast<const h8*>(wl + ZM_R1 + 3 * 2048 + 1024);
        fh[0] = *reinterpret_cast<const h8*>(img + a6);
        fl[0] = *reinterpret_cast<const h8*>(img + a6 + CS_PLANE);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int cur = m & 1, nxt = cur ^ 1;
            if (m < 3) {
                fh[nxt] = *reinterpret_cast<const h8*>(img + a6 + (m + 1) * PSTEP);
                fl[nxt] = *reinterpret_cast<const h8*>(img + a6 + (m + 1) * PSTEP + CS_PLANE);
            }
            __builtin_amdgcn_sched_barrier(0);
            hi[m] = mf(fh[cur], w6h, hi[m]);
            lo[m] = mf(fh[cur], w6l, lo[m]);
            lo[m] = mf(fl[cur], w6h, lo[m]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue: R0 of the first chunk, image of the first item
    if (tid < 384) *reinterpret_cast<h8*>(lds + ZM_R0 + tid * 16) = wsrc[wsrc0];
    Stage st;
    stage_load(st, b_first, 0);
    stage_store(st, b_first, 0, lds + ZM_IMG);
    lds_barrier();

    const int col = lane & 15;
    const int cob = blockIdx.y * 16;                                 // this workgroup's 16 couts of the layer (EPI 0: up to two blocks, grid.y)
    int item = 0;                                                    // parity of the image buffer
    for (int box = b_first; box < b_last; ++box) {
#pragma unroll
        for (int m = 0; m < 4; ++m) { hi[m] = (f32x4){0.f, 0.f, 0.f, 0.f}; lo[m] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        for (int ca = 0; ca < nC; ++ca, item ^= 1) {
            unsigned char* img = lds + ZM_IMG + item * CS_BUF;
            unsigned char* other = lds + ZM_IMG + (item ^ 1) * CS_BUF;
            // the item after this one (the very last item names itself: its loads are harmless and nothing branches around them)
            int nbox = box, nca = ca + 1;
            if (nca == nC) {
                nca = 0;
                nbox = box + 1;
                if (nbox == b_last) { nbox = box; nca = ca; }
            }
            const h8 w1 = (wsrc + (size_t)ca * wstride)[(unsigned)wsrc1];     // R1 of this chunk: lands under pass 0
            __builtin_amdgcn_sched_barrier(0);
            stage_load(st, nbox, nca);
            __builtin_amdgcn_sched_barrier(0);
            half_pass(img, std::integral_constant<int, 0>{});
            *reinterpret_cast<h8*>(lds + ZM_R1 + tid * 16) = w1;
            lds_barrier();                                            // MID: R1 in place, R0 free
            const h8 w0 = (wsrc + (size_t)nca * wstride)[(unsigned)wsrc0];    // R0 of the next chunk: lands under pass 1
            __builtin_amdgcn_sched_barrier(0);
            half_pass(img, std::integral_constant<int, 1>{});
            kstep6(img);
            const bool last = ca + 1 == nC;
            int n0, z0, y0, x0;
            box_origin(box, n0, z0, y0, x0);
            double fsm = 0.0, fsq = 0.0, psm = 0.0, psq = 0.0;       // this lane's share of the box's statistics (full output / pooled output)
            if (last) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) hi[m][r] = fmaxf(fmaf(lo[m][r], 1.0f / CS_LO, hi[m][r]), a.floor);      // the box's output values
            }
            if (EPI == 0 && last) {
                // outputs straight from the accumulators: D tile col = lane & 15 = cout, rows 4 (lane >> 4) + r = voxel x = i & 7, y = 2 yq + (i >> 3).
                // The store addresses depend on the box only, and hipcc hoists them out of the chunk loop as 64-bit pairs (six pairs held -- and spilled --
                // across every k-step of the box): an opaque copy of the lane's coordinates keeps them here
                int co_g = cob + col, kq = lane >> 4;
                asm volatile("" : "+v"(co_g), "+v"(kq));
                if (a.pool_mode != 2) {
                    if (co_g < a.cout) {
                        float* o = a.out + (size_t)n0 * a.cout * vol;                 // uniform; a sample is < 2^32 bytes (<= 32 couts x 128^3)
                        const unsigned o0 = (unsigned)co_g * (unsigned)vol + (unsigned)((z0 + 4 * zh) * edge + y0 + 2 * yq + (kq >> 1)) * (unsigned)edge + (unsigned)(x0 + 4 * (kq & 1));
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            *reinterpret_cast<float4*>(o + (o0 + (unsigned)(m * edge * edge))) = make_float4(hi[m][0], hi[m][1], hi[m][2], hi[m][3]);
                    }
                    if (a.stats) {
#pragma unroll
                        for (int m = 0; m < 4; ++m)
#pragma unroll
                            for (int r = 0; r < 4; ++r) { const double v = (double)fmaxf(hi[m][r], 0.f); fsm += v; fsq += v * v; }
                    }
                }
                if (a.pool_mode) {                                    // fused MaxPool3d(2): z pair = planes (2 zp, 2 zp + 1), x pairs in r, y pair in lane ^ 32
                    const int hedge = edge >> 1;
#pragma unroll
                    for (int zp = 0; zp < 2; ++zp) {
                        const f32x4 u = hi[2 * zp], v = hi[2 * zp + 1];
                        float p0 = fmaxf(fmaxf(u[0], u[1]), fmaxf(v[0], v[1]));
                        float p1 = fmaxf(fmaxf(u[2], u[3]), fmaxf(v[2], v[3]));
                        p0 = fmaxf(p0, __shfl_xor(p0, 32, 64));
                        p1 = fmaxf(p1, __shfl_xor(p1, 32, 64));
                        p0 = fmaxf(p0, 0.f);                          // max and ReLU commute
                        p1 = fmaxf(p1, 0.f);
                        if (kq < 2) {
                            psm += (double)p0 + (double)p1;
                            psq += (double)p0 * (double)p0 + (double)p1 * (double)p1;
                            if (co_g < a.cout)
                                *reinterpret_cast<float2*>(a.pool_out + (size_t)n0 * a.cout * ((size_t)hedge * hedge * hedge) +
                                                           ((unsigned)co_g * (unsigned)(hedge * hedge * hedge) +
                                                            (unsigned)(((z0 >> 1) + 2 * zh + zp) * hedge + (y0 >> 1) + yq) * (unsigned)hedge + (unsigned)((x0 >> 1) + 2 * kq))) = make_float2(p0, p1);
                        }
                    }
                }
            }
            stage_store(st, nbox, nca, other);
            if (tid < 384) *reinterpret_cast<h8*>(lds + ZM_R0 + tid * 16) = w0;
            lds_barrier();                                            // END: the next item's image and R0 in place; this image and R1 free
            if (!last) continue;
            if constexpr (EPI == 0) {
                if (a.stats || a.pool_stats) {
                    // statistics of the box: lane -> lane groups -> waves through the dead image, fixed order, float64
                    double2* red = reinterpret_cast<double2*>(img);
                    if (a.stats) {
                        fsm += __shfl_xor(fsm, 16, 64); fsq += __shfl_xor(fsq, 16, 64);
                        fsm += __shfl_xor(fsm, 32, 64); fsq += __shfl_xor(fsq, 32, 64);
                        if (lane < 16) red[wave * 16 + lane] = make_double2(fsm, fsq);
                    }
                    if (a.pool_stats) {
                        psm += __shfl_xor(psm, 16, 64); psq += __shfl_xor(psq, 16, 64);      // the two x halves (lane groups 0 and 1)
                        if (lane < 16) red[128 + wave * 16 + lane] = make_double2(psm, psq);
                    }
                    lds_barrier();
                    if (tid < 32) {
                        const int which = tid >> 4, co = tid & 15;
                        double2* dst = which ? a.pool_stats : a.stats;
                        if (dst && cob + co < a.cout) {
                            double s1 = 0.0, s2 = 0.0;
#pragma unroll
                            for (int w = 0; w < 8; ++w) { const double2 v = red[which * 128 + w * 16 + co]; s1 += v.x; s2 += v.y; }
                            dst[((size_t)n0 * a.cout + cob + co) * a.stats_tiles + (box & ((1 << (3 * tsh)) - 1))] = make_double2(s1, s2);
                        }
                    }
                }
            } else {
                // ReLU'd tile [couts][8^3] in the dead image
                float* e = reinterpret_cast<float*>(img);
                {
                    int lv = lane;
                    asm volatile("" : "+v"(lv));
                    const int colv = lv & 15, kqv = lv >> 4;
                    float* ew = e + colv * ZM_TILE_STRIDE + (4 * zh * 8 + 2 * yq + (kqv >> 1)) * 8 + 4 * (kqv & 1);
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) ew[m * 64 + r] = hi[m][r];
                }
                lds_barrier();
                if constexpr (EPI == 1) {
                    // pointwise head: per voxel the channel sum in the order of rf_conv1x1_tanh (bias first, channels ascending: the same bits)
                    int tv = tid;
                    asm volatile("" : "+v"(tv));                      // (keeps the output address out of the chunk loop's preheader, see EPI 0)
                    float accp = po.pw_b[0];
                    for (int c = 0; c < a.cout; ++c) accp = fmaf(e[c * ZM_TILE_STRIDE + tv], po.pw_w[c], accp);
                    const int z = tv >> 6, y = (tv >> 3) & 7, x = tv & 7;
                    (po.pw_out + (size_t)n0 * vol)[(unsigned)(((z0 + z) * edge + (y0 + y)) * edge + x0 + x)] = (tanhf(accp) + po.post_add) * po.post_mul;
                } else {
                    // pre-split output (whole 8^3 samples): statistics of the sample -> the next layer's GroupNorm triples -> normalise, split, slots
                    // (the arithmetic of k_conv3_split's pre-split epilogue, conv3d_split.hip)
                    double2* chst = reinterpret_cast<double2*>(lds + ZM_SCRATCH);
                    float4* trip = reinterpret_cast<float4*>(lds + ZM_SCRATCH + 256);
                    int tv = tid;
                    asm volatile("" : "+v"(tv));                      // (keeps the addresses below out of the chunk loop's preheader, see EPI 0)
                    {
                        const int co = tv >> 5, part = tv & 31;       // 32 threads per cout, 16 values each, then a butterfly (fixed order)
                        double sm = 0.0, sq = 0.0;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const float v = e[co * ZM_TILE_STRIDE + part + 32 * i];
                            sm += (double)v; sq += (double)v * v;
                        }
#pragma unroll
                        for (int msk = 1; msk < 32; msk <<= 1) { sm += __shfl_xor(sm, msk, 64); sq += __shfl_xor(sq, msk, 64); }
                        if (part == 0) {
                            chst[co] = make_double2(sm, sq);
                            if (a.stats && co < a.cout) a.stats[(size_t)n0 * a.cout + co] = make_double2(sm, sq);
                        }
                    }
                    lds_barrier();
                    if (tv < a.cout) {                                // as rf_gn_from_stats: group sums in channel order, float64
                        const int cpg = a.cout / po.groups, c0 = (tv / cpg) * cpg;
                        double sm = 0.0, sq = 0.0;
                        for (int c = c0; c < c0 + cpg; ++c) { sm += chst[c].x; sq += chst[c].y; }
                        const double count = (double)cpg * 512.0, mean = sm / count;
                        double var = sq / count - mean * mean;
                        if (var < 0.0) var = 0.0;
                        trip[tv] = gn_affine(mean, 1.0 / sqrt(var + (double)po.eps), po.gamma[tv], po.beta[tv]);
                    }
                    lds_barrier();
                    h8* __restrict__ o = po.out + (size_t)n0 * (a.cout >> 3) * 2 * 512 + (unsigned)tv;
                    for (int sg = 0; sg < (a.cout >> 3); ++sg) {
                        float y[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float4 t4 = trip[sg * 8 + j];
                            y[j] = fmaf(e[(sg * 8 + j) * ZM_TILE_STRIDE + tv] - t4.x, t4.y, t4.z);
                        }
                        h8 h, l;
                        cs_split8(y, h, l);
                        o[(size_t)sg * 2 * 512] = h;
                        o[(size_t)sg * 2 * 512 + 512] = l;
                    }
                }
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------------------------- host
// Taken by rf_conv3d_split_pre_k3_relu (conv3d_split.hip) for the shapes this kernel is built for.
bool rf_split_zc_takes(int cin, int n, int edge, int cout) {
    return cin == 8 && edge == 16 && cout > 0 && cout <= 16 && n >= 512;
}

int rf_split_zc_launch(const ConvArgs& a, const SplitPreOut& po, hipStream_t stream) {
    const bool full = a.pool_mode != 2;
    auto kern = full ? k_conv3_split_zc<true> : k_conv3_split_zc<false>;
    static RfLdsOptIn opt[2];
    if (int rc = opt[full].ensure(reinterpret_cast<const void*>(kern), ZC_LDS_BYTES, "rf_conv3d_split_pre_k3_relu")) return rc;
    const int wgs = a.n < rf_persistent_wgs() ? a.n : rf_persistent_wgs();      // two workgroups per CU, whole samples each
    const int per = (a.n + wgs - 1) / wgs;
    hipLaunchKernelGGL(kern, dim3((unsigned)((a.n + per - 1) / per)), dim3(512), ZC_LDS_BYTES, stream, a, po, per);
    RF_CHECK_LAUNCH("rf_conv3d_split_pre_k3_relu");
    return RF_OK;
}

// multi-chunk persistent form: up to 32 couts (16 per workgroup: the full-output form runs a 17..32-cout layer as two cout blocks on grid.y, each staging
// the boxes for itself -- with the operand reuse of the z-columns that is no more LDS traffic than one two-n-block workgroup of the box kernel and it keeps
// four waves per SIMD; the other epilogues take <= 16 couts), cin in whole chunks (>= 2), edge a power of two >= 8, enough boxes
bool rf_split_zcm_takes(int cin, int n, int edge, int cout) {
    const long long boxes = (long long)n * (edge / 8) * (edge / 8) * (edge / 8);
    return cin >= 16 && cin % 8 == 0 && cout > 0 && cout <= 32 && edge >= 8 && edge <= 128 && (edge & (edge - 1)) == 0 && boxes >= 2048;
}

template <bool PRE, int EPI, bool CH8 = false>
static int zcm_launch(const ConvArgs& a, const SplitPreOut& po, hipStream_t stream, const char* who) {
    auto kern = k_conv3_split_zcm<PRE, EPI, CH8>;
    static RfLdsOptIn opt;
    if (int rc = opt.ensure(reinterpret_cast<const void*>(kern), ZM_LDS_BYTES, who)) return rc;
    const long long boxes = (long long)a.n * (a.edge / 8) * (a.edge / 8) * (a.edge / 8);
    const int wgs = rf_persistent_wgs();                              // two workgroups per CU
    const unsigned cob_blocks = (unsigned)(a.cout16 / 16);
    const int per = (int)((boxes * cob_blocks + wgs - 1) / wgs);
    hipLaunchKernelGGL(kern, dim3((unsigned)((boxes + per - 1) / per), cob_blocks), dim3(512), ZM_LDS_BYTES, stream, a, po, per, (int)boxes);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rf_set_error("%s: launch failed: %s", who, hipGetErrorString(e)); return RF_E_LAUNCH; }
    return RF_OK;
}

// pre: the input is a pre-split tensor; po.pw_out set: the pointwise-head epilogue, else the full output (+ a.stats)
int rf_split_zcm_launch(const ConvArgs& a, const SplitPreOut& po, bool pre, hipStream_t stream, const char* who) {
    if (po.pw_out) return pre ? zcm_launch<true, 1>(a, po, stream, who) : zcm_launch<false, 1>(a, po, stream, who);
    if (po.out) return pre ? zcm_launch<true, 2>(a, po, stream, who) : zcm_launch<false, 2>(a, po, stream, who);
    return pre ? zcm_launch<true, 0>(a, po, stream, who) : zcm_launch<false, 0>(a, po, stream, who);
}

// the pointwise-head epilogue on a channel-interleaved fp32 input (rf_conv3d_split_k3_gn_relu_pointwise_tanh_ch8)
int rf_split_zcm_launch_ch8_pointwise(const ConvArgs& a, const SplitPreOut& po, hipStream_t stream, const char* who) {
    return zcm_launch<false, 1, true>(a, po, stream, who);
}
