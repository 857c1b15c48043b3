// 3x3x3 conv + GroupNorm apply + ReLU on WHOLE 4^3 / 2^3 volumes, position-major: the third tiling of
// rf_conv3d_k3_gn_relu (reference model/unet.py:19-76, the two coarsest levels of the retrieval backbone).
//
// In a 4^3 (2^3) volume 42 % (70 %) of the 27 taps of an output voxel read zero padding.  The generic kernel
// (conv3d_mfma.hip) builds MFMA m-blocks from 16 voxels of one sample and can only leave out taps that are padding for the
// whole m-block (the z border).  Here an m-block is ONE VOXEL POSITION of 16 DIFFERENT SAMPLES: whether tap (dz,dy,dx) of
// position (z,y,x) is padding is then a property of the m-block, known at compile time, and every padding tap is simply
// not issued -- no halo is staged either.  The MFMAs left out would have added exact zeros: the result is bit for bit
// the one of the generic kernel.
//
//   4^3: a workgroup of 8 waves owns 16 samples; wave w owns the 8 positions z = w >> 1, y in {2*(w&1), 2*(w&1)+1}, x = 0..3
//        (m-block mb: y = 2*(w&1) + (mb >> 2), x = mb & 3).  Which taps exist depends on the wave only through the class of
//        z (first / interior / last slice) and the y half: 6 variants of the whole K loop + epilogue, a wave picks one.
//   2^3: a workgroup owns 128 samples = 8 groups of 16; wave w owns group w, m-block mb = position mb (one variant).
//
// LDS: input chunk [position][4 ch][16 samples] (lane -> 16*k + sample: conflict-free operand reads, the tap is an
// immediate offset), weight slabs [27][4][NB*16] by LDS-DMA, double buffered, as conv3d_mfma.hip.  Epilogue: accumulators
// -> LDS -> contiguous float4 rows; GroupNorm statistics of the output for the next layer (float64, fixed order).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((address_space(1))) const void* rf_gptr;
typedef __attribute__((address_space(3))) void* rf_lptr;

struct SmallArgs {
    const float* src;
    const float4* affine;   // GroupNorm per (sample, input channel): (center, scale, shift, -) -> y = (x - center) * scale + shift
    const float* wp;       // [27][cin4][cout16] (rf_conv3_pack_weight)
    float* out;
    int cin, n, cout, cin4, cout16;
    double2* stats;        // optional [n][cout][1]
    // decoder form (4^3 only, template flag UP): `src` / `cin` are the skip source (may be 0 channels), `src1` [n][c1][2^3] the
    // low-res source convolved with the parity-split, pre-summed taps of conv3d_up.hip; wp / wp1 = the two parts of the
    // rf_conv3_up_pack_weight image; GroupNorm scale / shift are indexed over the cin + c1 concatenated channels
    const float* src1;
    const float* wp1;      // [c1_8/8][8 parities][8 taps][8 ch][cout16]
    int c1, c1_8;
    float* pool_out;       // optional (4^3): MaxPool3d(2) of the output [n][cout][2^3] and its statistics [n][cout][1]; `out` may then be NULL
    double2* pool_stats;
};

template <int E, int NB, bool UP = false>
struct SmallTile {
    static constexpr int P = E * E * E;                       // positions per sample
    static constexpr int NW = 8, NT = 512, MB = 8;
    static constexpr int GROUPS = (NW * MB) / P;              // 16-sample groups per workgroup: 1 (4^3) or 8 (2^3)
    static constexpr int SAMPLES = 16 * GROUPS;
    static constexpr int NCO = NB * 16;
    static constexpr int XS = GROUPS * P * 64;                // [group][position][4 ch][16 samples]
    static constexpr int WSLAB = 27 * 4 * NCO;
    static constexpr int WSLAB_PAD = (WSLAB + 255) / 256 * 256;
    static constexpr int EROW = P + 1;                        // epilogue tile [16 cout][SAMPLES][P + 1]
    static constexpr int EPI = 16 * SAMPLES * EROW;
    static constexpr int MAIN = XS + 2 * WSLAB_PAD;
    static constexpr int XLOW = 2 * 8 * 64;                   // decoder form: two low-res chunks [2^3 positions][4 ch][16 samples]
    static constexpr int BSLAB = 8 * 8 * 4 * NCO;             // ... and weight slabs [8 parities][8 taps][4 ch][NCO], double buffered
    static constexpr int MAINB = UP ? XLOW + 2 * BSLAB : 0;
    static constexpr int MAXAB = MAIN > MAINB ? MAIN : MAINB;
    static constexpr size_t LDS_BYTES = (size_t)(MAXAB > EPI ? MAXAB : EPI) * sizeof(float);
    static_assert(!UP || E == 4, "decoder form: 4^3 volumes");
    static_assert(LDS_BYTES <= 81920, "two workgroups per CU");
    static_assert(E == 4 || E == 2, "whole 4^3 / 2^3 volumes");
};

// position of m-block mb of wave `wave` (4^3) / of any wave (2^3)
template <int E>
__device__ __forceinline__ constexpr int small_y(int yh, int mb) { return E == 4 ? 2 * yh + (mb >> 2) : (mb >> 1) & 1; }
template <int E>
__device__ __forceinline__ constexpr int small_x(int mb) { return E == 4 ? (mb & 3) : (mb & 1); }

template <int E, int NB, bool UP>
__global__ __launch_bounds__(512, 4) void k_conv3_small(SmallArgs a) {
    using T = SmallTile<E, NB, UP>;
    constexpr int P = T::P, NCO = T::NCO, NT = T::NT, MB = T::MB, SAMPLES = T::SAMPLES, EROW = T::EROW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* wsb = smem + T::XS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * SAMPLES, cob = blockIdx.y * NCO;
    const int kq = lane >> 4, li = lane & 15;
    // 4^3: z slice and y half of this wave; 2^3: sample group of this wave
    const int wz = E == 4 ? (wave >> 1) : 0, wyh = E == 4 ? (wave & 1) : 0, wgrp = E == 4 ? 0 : wave;

    constexpr int ROT = NCO >= 32 ? 16 : 0;
    int boff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) boff[nb] = kq * NCO + ((nb * 16 + li + ROT * (kq & 1)) % NCO);
    // operand base: [group][position][k][sample]; the position of the input voxel is added as an immediate (+ z of the wave)
    const int abase = (wgrp * P + (E == 4 ? wz * 16 : 0)) * 64 + kq * 16 + li;

    // Weight slab DMA: 1-KiB pieces of RPP slab rows [tap][channel] x NCO.  Per-lane index arithmetic once (it runs on the VALU port
    // the MFMAs issue through), per chunk and piece a scalar base -- see conv3d_mfma.hip.
    constexpr int LPR = NCO / 4, RPP = 64 / LPR;                    // lanes per slab row; slab rows per piece (4, 8, 16)
    constexpr int NPIECE = (27 * 4 + RPP - 1) / RPP;
    const int wr = lane / LPR, wslot = (lane % LPR) * 4;
    const int wcol = (wslot + NCO - ROT * (wr & 1)) % NCO;
    int wco = cob + wcol;
    if (wco >= a.cout16) wco = wcol % a.cout16;
    const unsigned wlane = 4u * (unsigned)(((wr >> 2) * a.cin4 + (wr & 3)) * a.cout16 + wco);        // bytes
    auto dma_weights = [&](int cbase, int buf) {
        float* dst = wsb + buf * T::WSLAB_PAD;
        unsigned off = wlane;                                       // opaque: SGPR base + 32-bit lane offset addressing
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int i = 0; i < (NPIECE + 7) / 8; ++i) {
            const int q = wave + i * 8;
            if (q < NPIECE) {
                const int tap0 = q * (RPP / 4);
                const char* src = reinterpret_cast<const char*>(a.wp + ((size_t)tap0 * a.cin4 + cbase) * a.cout16);
                if (RPP == 4 || tap0 + (wr >> 2) < 27)              // rows past tap 26 of the last piece are never read
                    __builtin_amdgcn_global_load_lds((rf_gptr)(src + off), (rf_lptr)(dst + q * 256), 16, 0, 0);
            }
        }
    };

    // ---- input chunk: item = (sample, channel, run of 8 positions); 8 floats per item, GroupNorm applied at commit
    constexpr int RUNS = P / 8;                                     // 8 (4^3) or 1 (2^3)
    constexpr int ITEMS = SAMPLES * 4 * RUNS;                       // 512 either way
    static_assert(ITEMS == NT, "one staging item per thread");
    float xraw[8];
    float xce = 0.f, xsc = 0.f, xsh = 0.f;
    // thread -> (sample, k) fastest so that a wave's LDS writes of one position are conflict-free
    const int it_s = tid % SAMPLES, it_k = (tid / SAMPLES) % 4, it_run = tid / (SAMPLES * 4);
    auto issue_rows = [&](int cbase) {
        const int nn = n0 + it_s, ci = cbase + it_k;
        if (nn < a.n && ci < a.cin) {
            const size_t si = (size_t)nn * a.cin + ci;
            const size_t gi = (size_t)nn * (a.cin + (UP ? a.c1 : 0)) + ci;
            { const float4 af = a.affine[gi]; xce = af.x; xsc = af.y; xsh = af.z; }
            const float4* row = reinterpret_cast<const float4*>(a.src + si * P + it_run * 8);
            const float4 t0 = row[0], t1 = row[1];
            xraw[0] = t0.x; xraw[1] = t0.y; xraw[2] = t0.z; xraw[3] = t0.w;
            xraw[4] = t1.x; xraw[5] = t1.y; xraw[6] = t1.z; xraw[7] = t1.w;
        }
    };
    auto commit_rows = [&](int cbase) {
        const int nn = n0 + it_s, ci = cbase + it_k;
        const bool ok = nn < a.n && ci < a.cin;
        float* dst = xs + ((it_s >> 4) * P + it_run * 8) * 64 + it_k * 16 + (it_s & 15);
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j * 64] = ok ? fmaf(xraw[j] - xce, xsc, xsh) : 0.f;
    };

    // Everything below is instantiated per variant of this wave (ZC: 0 = first z slice, 1 = interior, 2 = last; YH: y half)
    auto run = [&](auto zc_c, auto yh_c) {
        constexpr int ZC = decltype(zc_c)::value, YH = decltype(yh_c)::value;
        f32x4 acc[MB][NB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

        if (a.cin > 0) {
            dma_weights(0, 0);
            issue_rows(0);
            commit_rows(0);
        }
        __syncthreads();

        int buf = 0;
        for (int cbase = 0; cbase < a.cin; cbase += 4) {
            const bool more = cbase + 4 < a.cin;
            if (more) {
                issue_rows(cbase + 4);
                dma_weights(cbase + 4, buf ^ 1);
            }
            const float* ws = wsb + buf * T::WSLAB_PAD;
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
                // z validity of this tap: the same for all 8 m-blocks of the wave (4^3), per m-block for 2^3
                if (E == 4 && ((ZC == 0 && dz == 0) || (ZC == 2 && dz == 2))) continue;
                asm volatile("" ::: "memory");                       // keep the operand reads of later taps out of this one
                float bv[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bv[nb] = ws[boff[nb] + t * 4 * NCO];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int z = E == 4 ? 0 : (mb >> 2), y = small_y<E>(YH, mb), x = small_x<E>(mb);
                    const int uy = y + dy - 1, ux = x + dx - 1, uz = z + dz - 1;        // input voxel (z relative to the wave's slice at 4^3)
                    if (uy < 0 || uy >= E || ux < 0 || ux >= E) continue;              // compile-time: padding tap
                    if (E == 2 && (uz < 0 || uz >= E)) continue;
                    const int upos = (E == 4 ? (dz - 1) * 16 : uz * 4) + uy * E + ux;  // position offset from the operand base
                    const float av = xs[abase + upos * 64];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[nb], acc[mb][nb], 0, 0, 0);
                }
            }
            __syncthreads();
            if (more) commit_rows(cbase + 4);
            __syncthreads();
            buf ^= 1;
        }

        // ---- decoder form, phase B: the c1 channels of the low-res (2^3) source.  Output position (z,y,x) has parity
        // p = (z&1, y&1, x&1) and lattice cell (Z,Y,X) = (z,y,x) >> 1; its low-res tap (tz,ty,tx) reads low-res voxel
        // (Z + tz - 1 + pz, ...) with the pre-summed weight W'[parity][tap] -- or zero padding, which is not issued.
        if constexpr (UP) {
            float* xlow = smem;                                      // [2][8 positions][4 ch][16 samples]
            float* bsl = smem + T::XLOW;                             // [2][(parity*8 + tap)*4 + k][NCO]
            constexpr int BROT = NCO == 32 ? 16 : 0;                 // rows k, k+2 of a B read would share banks at NCO = 32
            constexpr int BF4 = T::BSLAB / 4, BPIECE = BF4 / 64;
            // slab row r = (parity*8 + tap)*4 + k; piece q = rows [q RPP, (q+1) RPP): lane part once, scalar base per chunk and piece
            const int bw = lane / LPR;                               // row of this lane inside a piece
            const int bcol = ((lane % LPR) * 4 + NCO - BROT * ((bw >> 1) & 1)) % NCO;
            int bco = cob + bcol;
            if (bco >= a.cout16) bco = bcol % a.cout16;
            const unsigned blane = 4u * (unsigned)(((bw >> 2) * 8 + (bw & 3)) * a.cout16 + bco);          // bytes
            auto dma_b = [&](int c4, int bufb) {                     // c4: 4-channel chunk of c1
                float* dst = bsl + bufb * T::BSLAB;
                unsigned off = blane;
                asm volatile("" : "+v"(off));
#pragma unroll
                for (int i = 0; i < (BPIECE + 7) / 8; ++i) {
                    const int q = wave + i * 8;
                    if (q < BPIECE) {
                        const char* src = reinterpret_cast<const char*>(a.wp1 + ((((size_t)(c4 >> 1) * 64 + q * (RPP / 4)) * 8) + (c4 & 1) * 4) * a.cout16);
                        __builtin_amdgcn_global_load_lds((rf_gptr)(src + off), (rf_lptr)(dst + q * 256), 16, 0, 0);
                    }
                }
            };
            float lraw[8];
            float lce = 0.f, lsc = 0.f, lsh = 0.f;
            const int ls = tid & 15, lk = (tid >> 4) & 3;            // threads 0..63 stage the low-res chunk
            auto issue_low = [&](int c4) {
                const int nn = n0 + ls, ci = c4 * 4 + lk;
                if (tid < 64 && nn < a.n && ci < a.c1) {
                    const size_t gi = (size_t)nn * (a.cin + a.c1) + a.cin + ci;
                    { const float4 af = a.affine[gi]; lce = af.x; lsc = af.y; lsh = af.z; }
                    const float4* row = reinterpret_cast<const float4*>(a.src1 + ((size_t)nn * a.c1 + ci) * 8);
                    const float4 t0 = row[0], t1 = row[1];
                    lraw[0] = t0.x; lraw[1] = t0.y; lraw[2] = t0.z; lraw[3] = t0.w;
                    lraw[4] = t1.x; lraw[5] = t1.y; lraw[6] = t1.z; lraw[7] = t1.w;
                }
            };
            auto commit_low = [&](int c4, int bufb) {
                if (tid < 64) {
                    const int nn = n0 + ls, ci = c4 * 4 + lk;
                    const bool ok = nn < a.n && ci < a.c1;
                    float* dst = xlow + bufb * 512 + lk * 16 + ls;
#pragma unroll
                    for (int j = 0; j < 8; ++j) dst[j * 64] = ok ? fmaf(lraw[j] - lce, lsc, lsh) : 0.f;
                }
            };
            const int nchunk = a.c1_8 >> 2;
            int bboff[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bboff[nb] = kq * NCO + ((nb * 16 + li + BROT * ((kq >> 1) & 1)) % NCO);
            const int pz = wz & 1;                                   // z parity of this wave's slice (run time inside ZC == 1)
            dma_b(0, 0);
            issue_low(0);
            commit_low(0, 0);
            __syncthreads();
            int bufb = 0;
            for (int c4 = 0; c4 < nchunk; ++c4) {
                const bool more = c4 + 1 < nchunk;
                if (more) {
                    issue_low(c4 + 1);
                    dma_b(c4 + 1, bufb ^ 1);
                }
                const float* xl = xlow + bufb * 512 + kq * 16 + li;
                const float* wb = bsl + bufb * T::BSLAB + pz * (4 * 8 * 4 * NCO);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int tz = t >> 2, ty = (t >> 1) & 1, tx = t & 1;
                    // low-res z of this tap: first slice (Z=0,pz=0): tz=0 is padding, tz=1 -> 0; last slice (Z=1,pz=1): tz=0 -> 1, tz=1
                    // is padding; interior slices (Z=0,pz=1 / Z=1,pz=0): tz -> tz
                    if ((ZC == 0 && tz == 0) || (ZC == 2 && tz == 1)) continue;
                    const int uz = ZC == 0 ? 0 : (ZC == 2 ? 1 : tz);
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        const int y = small_y<E>(YH, mb), x = small_x<E>(mb);
                        const int py = y & 1, px = x & 1;
                        const int uy = (y >> 1) + ty - 1 + py, ux = (x >> 1) + tx - 1 + px;
                        if (uy < 0 || uy > 1 || ux < 0 || ux > 1) continue;              // compile-time: padding tap
                        const float av = xl[(uz * 4 + uy * 2 + ux) * 64];
                        const int brow = ((py * 2 + px) * 8 + t) * 4;                    // + pz*4 parities through wb
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wb[brow * NCO + bboff[nb]], acc[mb][nb], 0, 0, 0);
                    }
                }
                if (more) commit_low(c4 + 1, bufb ^ 1);
                __syncthreads();
                bufb ^= 1;
            }
        }

        // ---- epilogue: ReLU'd accumulators -> LDS [16 cout][sample][P + 1] -> contiguous float4 rows; statistics
        float* eb = smem;
        double* red = reinterpret_cast<double*>(smem);               // [8 waves][16 samples][16 cout][2] (4^3 only), after the stores
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int pos = E == 4 ? wz * 16 + small_y<E>(YH, mb) * 4 + small_x<E>(mb) : mb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int smp = wgrp * 16 + kq * 4 + r;          // D rows of this lane: samples 4*kq + r of the group
                    eb[(li * SAMPLES + smp) * EROW + pos] = fmaxf(acc[mb][nb][r], 0.f);
                }
            }
            __syncthreads();
            for (int q = tid; q < 16 * SAMPLES * (P / 4); q += NT) {
                const int p4 = q % (P / 4), smp = (q / (P / 4)) % SAMPLES, col = q / ((P / 4) * SAMPLES);
                const int co = cob + nb * 16 + col, nn = n0 + smp;
                if (co < a.cout && nn < a.n && a.out) {
                    const float* e = eb + (col * SAMPLES + smp) * EROW + p4 * 4;
                    *reinterpret_cast<float4*>(a.out + ((size_t)nn * a.cout + co) * P + p4 * 4) = make_float4(e[0], e[1], e[2], e[3]);
                }
            }
            if (E == 4 && a.pool_out) {
                // fused MaxPool3d(2) (4^3 -> 2^3) straight from the tile, and the pooled tensor's GroupNorm statistics: one thread per
                // (cout, sample), fixed order
                if (tid < 256) {
                    const int smp = tid >> 4, col = tid & 15;
                    const int co = cob + nb * 16 + col, nn = n0 + smp;
                    if (co < a.cout && nn < a.n) {
                        const float* e = eb + (col * SAMPLES + smp) * EROW;
                        float pv[8];
                        double sm = 0.0, sq = 0.0;
#pragma unroll
                        for (int cell = 0; cell < 8; ++cell) {
                            const int base = (cell >> 2) * 32 + ((cell >> 1) & 1) * 8 + (cell & 1) * 2;      // (2Z, 2Y, 2X)
                            float m = fmaxf(fmaxf(e[base], e[base + 1]), fmaxf(e[base + 4], e[base + 5]));
                            m = fmaxf(m, fmaxf(fmaxf(e[base + 16], e[base + 17]), fmaxf(e[base + 20], e[base + 21])));
                            pv[cell] = m;
                            sm += (double)m; sq += (double)m * (double)m;
                        }
                        float4* po = reinterpret_cast<float4*>(a.pool_out + ((size_t)nn * a.cout + co) * 8);
                        po[0] = make_float4(pv[0], pv[1], pv[2], pv[3]);
                        po[1] = make_float4(pv[4], pv[5], pv[6], pv[7]);
                        if (a.pool_stats) a.pool_stats[(size_t)nn * a.cout + co] = make_double2(sm, sq);
                    }
                }
            }
            __syncthreads();
            if (a.stats) {
                // per (sample, cout): sum over this wave's 8 positions in registers; 4^3: then over the 8 waves through LDS
                double sm[4], sq[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sm[r] = 0.0; sq[r] = 0.0;
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        const double v = (double)fmaxf(acc[mb][nb][r], 0.f);
                        sm[r] += v; sq[r] += v * v;
                    }
                }
                if (E == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = cob + nb * 16 + li, nn = n0 + wgrp * 16 + kq * 4 + r;
                        if (co < a.cout && nn < a.n) a.stats[(size_t)nn * a.cout + co] = make_double2(sm[r], sq[r]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        red[((wave * 16 + kq * 4 + r) * 16 + li) * 2] = sm[r];
                        red[((wave * 16 + kq * 4 + r) * 16 + li) * 2 + 1] = sq[r];
                    }
                    __syncthreads();
                    if (tid < 256) {
                        const int smp = tid >> 4, col = tid & 15;
                        const int co = cob + nb * 16 + col, nn = n0 + smp;
                        if (co < a.cout && nn < a.n) {
                            double s0 = 0.0, s1 = 0.0;
#pragma unroll
                            for (int w = 0; w < 8; ++w) {
                                s0 += red[((w * 16 + smp) * 16 + col) * 2];
                                s1 += red[((w * 16 + smp) * 16 + col) * 2 + 1];
                            }
                            a.stats[(size_t)nn * a.cout + co] = make_double2(s0, s1);
                        }
                    }
                    __syncthreads();
                }
            }
        }
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    if (E == 2) {
        run(I1{}, I0{});
    } else {
        const int zc = wz == 0 ? 0 : (wz == E - 1 ? 2 : 1);
        if (wyh == 0) {
            if (zc == 0) run(I0{}, I0{}); else if (zc == 1) run(I1{}, I0{}); else run(I2{}, I0{});
        } else {
            if (zc == 0) run(I0{}, I1{}); else if (zc == 1) run(I1{}, I1{}); else run(I2{}, I1{});
        }
    }
}

template <int E, int NB, bool UP = false>
static int launch_small(const SmallArgs& a, hipStream_t stream) {
    using T = SmallTile<E, NB, UP>;
    auto kern = k_conv3_small<E, NB, UP>;
    if (T::LDS_BYTES > 65536) {
        static RfLdsOptIn opt_in;
        if (int rc = opt_in.ensure(reinterpret_cast<const void*>(kern), (int)T::LDS_BYTES, "rf_conv3d_k3_gn_relu(small)")) return rc;
    }
    const unsigned gx = (unsigned)((a.n + T::SAMPLES - 1) / T::SAMPLES), gy = (unsigned)((a.cout16 + T::NCO - 1) / T::NCO);
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(T::NT), T::LDS_BYTES, stream, a);
    RF_CHECK_LAUNCH("rf_conv3d_k3_gn_relu(small)");
    return RF_OK;
}

// Takes whole 4^3 / 2^3 volumes with a single (full-resolution) source when there are enough samples to fill the chip.
bool rf_conv3_small_takes(int c0, int c1, int n, int edge, int cout) {
    if (c1 != 0 || c0 <= 0 || (edge != 4 && edge != 2)) return false;
    const long long gy = (rf_round_up(cout, 16) + 31) / 32;
    const long long wgs = (edge == 4 ? (n + 15) / 16 : (n + 127) / 128) * gy;
    return wgs >= 128;                                               // half a wave of workgroups per CU still beats the box tiling
}

int rf_conv3_small_launch(const float* src, int cin, int n, int edge, const float* gn_affine, const float* w_packed, int cout,
                          float* out, double* stats, void* stream, float* pool_out, double* pool_stats) {
    SmallArgs a;
    a.src = src; a.affine = reinterpret_cast<const float4*>(gn_affine); a.wp = w_packed; a.out = out;
    a.cin = cin; a.n = n; a.cout = cout; a.cin4 = rf_round_up(cin, 4); a.cout16 = rf_round_up(cout, 16);
    a.stats = reinterpret_cast<double2*>(stats);
    a.src1 = nullptr; a.wp1 = nullptr; a.c1 = 0; a.c1_8 = 0;
    a.pool_out = pool_out; a.pool_stats = reinterpret_cast<double2*>(pool_stats);
    hipStream_t s = (hipStream_t)stream;
    if (edge == 4) return a.cout16 <= 16 ? launch_small<4, 1>(a, s) : launch_small<4, 2>(a, s);
    return a.cout16 <= 16 ? launch_small<2, 1>(a, s) : launch_small<2, 2>(a, s);
}

// decoder form on whole 4^3 volumes (low-res source 2^3), weight image of rf_conv3_up_pack_weight
bool rf_conv3_small_up_takes(int c0, int c1, int n, int edge, int cout) {
    if (c1 <= 0 || c0 < 0 || edge != 4) return false;
    const long long gy = (rf_round_up(cout, 16) + 31) / 32;
    return (long long)((n + 15) / 16) * gy >= 256;
}

int rf_conv3_small_up_launch(const float* src0, int c0, const float* src1, int c1, int n, const float* gn_affine,
                             const float* w_up_packed, int cout, float* out, double* stats, void* stream) {
    SmallArgs a;
    a.src = src0; a.affine = reinterpret_cast<const float4*>(gn_affine); a.wp = w_up_packed; a.out = out;
    a.cin = c0; a.n = n; a.cout = cout; a.cin4 = rf_round_up(c0, 4); a.cout16 = rf_round_up(cout, 16);
    a.stats = reinterpret_cast<double2*>(stats);
    a.src1 = src1; a.c1 = c1; a.c1_8 = rf_round_up(c1, 8);
    a.pool_out = nullptr; a.pool_stats = nullptr;
    a.wp1 = w_up_packed + (size_t)27 * a.cin4 * a.cout16;
    hipStream_t s = (hipStream_t)stream;
    return a.cout16 <= 16 ? launch_small<4, 1, true>(a, s) : launch_small<4, 2, true>(a, s);
}
