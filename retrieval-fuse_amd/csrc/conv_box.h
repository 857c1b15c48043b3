// Shared by the box-tiled 3x3x3 convolution kernels (conv3d_mfma.hip: fp32 MFMA; conv3d_split.hip: split-operand F16 MFMA): launch
// arguments, the voxel order of a wave's accumulator tiles, the XCD-contiguous workgroup order and the EPILOGUE (ReLU + float4 stores,
// GroupNorm statistics of the output, fused MaxPool3d(2) + its statistics).  Both MFMA families deliver a 16x16 tile with the same
// lane map (col = lane & 15, rows 4*(lane >> 4) + r), so one epilogue serves both and their outputs / statistics have one layout.
#pragma once
#include "common.h"

struct ConvArgs {
    const float* src0;
    const float* src1;
    const float4* affine;   // GroupNorm per (sample, input channel): (center, scale, shift, -) -> y = (x - center) * scale + shift
    const float* wp;
    float* out;
    int c0, c1, n, edge, cout, cin4, cout16;
    double2* stats;    // optional [n][cout][stats_tiles] (sum, sum of squares) of the ReLU'd output, per workgroup tile
    int stats_tiles;
    // fused MaxPool3d(2) of the output (8^3 boxes only): pool_out [n][cout][(edge/2)^3] and its statistics
    // [n][cout][stats_tiles]; pool_mode 0 = off, 1 = write both, 2 = pooled only (`out` is not written: an encoder level
    // whose full-resolution output nobody reads, model/unet.py:500-507 with remove_n_final_layers)
    float* pool_out;
    double2* pool_stats;
    int pool_mode;
    float floor;       // output clamp from below: 0 = ReLU (the forward layers), -inf = none (rf_conv3d_k3_gn with relu = 0: the dgrad conv)
    // pre-split input of whole 8^3 samples in PARITY-MAJOR slot order (what rf_conv3d_up_split_presplit_pm writes: the slot of voxel (z, y, x) is
    // ((z & 1) 4 + (y & 1) 2 + (x & 1)) 64 + (z >> 1) 16 + (y >> 1) 4 + (x >> 1)); k_conv3_split_zcm<PRE> only
    int src_pm = 0;
};

// Voxel order inside the 8^3 box of the 8-wave x MB 4 tile: m = wave*64 + mb*16 + i -> (z, y, x) such that one lane's
// accumulators (mb = 0..3, rows r = 0..3 of its lane group) hold whole 2x2x2 pooling cells up to one lane exchange:
//   x = i & 7, y = 4*(wave & 1) + 2*(mb >> 1) + (i >> 3), z = 2*(wave >> 1) + (mb & 1)
// (z pairs in mb, x pairs in r, y pairs in lanes l / l^32).  Other tiles keep the plain row-major order.
template <int TZ, int TY, int TX, int NW, int MB>
struct BoxOrder {
    static constexpr bool POOLABLE = TZ == 8 && TY == 8 && TX == 8 && NW == 8 && MB == 4;
    __device__ static __forceinline__ void voxel(int wave, int mb, int i, int& s, int& z, int& y, int& x) {
        if (POOLABLE) {
            s = 0; x = i & 7; y = 4 * (wave & 1) + 2 * (mb >> 1) + (i >> 3); z = 2 * (wave >> 1) + (mb & 1);
        } else {
            const int m = wave * (MB * 16) + mb * 16 + i;
            x = m % TX; y = (m / TX) % TY; z = (m / (TX * TY)) % TZ; s = m / (TX * TY * TZ);
        }
    }
};

// Workgroups are handed to the 8 XCDs round-robin by linear id, each XCD with its own L2.  Tiles that split a sample
// (8-voxel = 32-byte row pieces of 64..512-byte rows) would then share every cache line across XCDs: partial-line writes
// that no L2 can merge and 128-byte fills for 32 bytes of use.  Remap so that XCD k walks a contiguous range of tiles.
__device__ __forceinline__ unsigned rf_xcd_contiguous(unsigned b, unsigned g) {
    const unsigned per = g >> 3, rem = g & 7u, k = b & 7u;
    return k * per + (k < rem ? k : rem) + (b >> 3);
}

// 512-voxel workgroup tiles (8^3 boxes) when that still gives the 256 CUs a few workgroups each; otherwise the fp32 kernel's 128-voxel tiles
static inline bool rf_conv_use_big(int n, int edge, int cout16) {
    const long long vox = (long long)n * edge * edge * edge;
    const long long wgs512 = (vox + 511) / 512 * ((cout16 + 63) / 64);
    return wgs512 >= 1024;
}

// acc[mb][nb]: the wave's MB x NB accumulator tiles (pre-ReLU); smem: the workgroup's LDS (>= LDS_BYTES, free to overwrite: the K loop
// ended on a barrier); (n0, z0, y0, x0): first sample / box origin; cob: first cout of the workgroup; lblock: linear tile index.
template <int TZ, int TY, int TX, int SPW, int NW, int MB, int NB, size_t LDS_BYTES>
__device__ __forceinline__ void conv_box_epilogue(const ConvArgs& a, f32x4 (&acc)[MB][NB], float* smem, const int tid, const int lane, const int wave,
                                                  const int n0, const int z0, const int y0, const int x0, const int cob, const unsigned lblock) {
    constexpr int NCO = NB * 16, NT = NW * 64;
    const int edge = a.edge;
    // ---- epilogue: ReLU, float4 stores (a lane holds 4 consecutive voxels of one cout)
    const size_t vol = (size_t)edge * edge * edge;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        int s, z, y, x;
        BoxOrder<TZ, TY, TX, NW, MB>::voxel(wave, mb, (lane >> 4) * 4, s, z, y, x);
        const int nn = n0 + s;
        size_t off;
        if (TX >= 4) {
            off = ((size_t)(z0 + z) * edge + (y0 + y)) * edge + (x0 + x);
        } else {
            off = (size_t)((wave * (MB * 16) + mb * 16 + (lane >> 4) * 4) % (TX * TY * TZ));   // tile == whole volume: voxel order is memory order
        }
        if (nn < a.n && a.pool_mode != 2) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int co = cob + nb * 16 + (lane & 15);
                if (co < a.cout) {
                    f32x4 v = acc[mb][nb];
                    float4 o = make_float4(fmaxf(v[0], a.floor), fmaxf(v[1], a.floor), fmaxf(v[2], a.floor), fmaxf(v[3], a.floor));
                    *reinterpret_cast<float4*>(a.out + ((size_t)nn * a.cout + co) * vol + off) = o;
                }
            }
        }
    }

    // ---- optional: GroupNorm statistics of the output for the NEXT layer, per (sample, cout), this workgroup's tile.
    // Fixed reduction order (registers -> lane groups -> waves through LDS), float64: deterministic, no atomics.
    if (a.stats) {
        constexpr int VOL = TZ * TY * TX;                       // voxels of one sample inside the tile
        constexpr int WV = MB * 16;                             // voxels per wave
        constexpr int SLOTS = VOL >= WV ? 1 : WV / VOL;         // sample slots per wave (2^3 volumes: several samples per wave)
        static_assert(VOL >= WV || VOL == 8, "sub-wave samples are 2^3 volumes");
        static_assert((size_t)NW * SLOTS * NCO * 2 * sizeof(double) <= LDS_BYTES, "stats scratch must fit the tile's LDS");
        double* red = reinterpret_cast<double*>(smem);          // [NW][SLOTS][NCO][2]; the K loop ended on a barrier
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (SLOTS == 1) {
                double sm = 0.0, sq = 0.0;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double v = (double)fmaxf(acc[mb][nb][r], 0.f);
                        sm += v; sq += v * v;
                    }
                sm += __shfl_xor(sm, 16, 64); sq += __shfl_xor(sq, 16, 64);
                sm += __shfl_xor(sm, 32, 64); sq += __shfl_xor(sq, 32, 64);
                if (lane < 16) {
                    red[((size_t)wave * NCO + nb * 16 + lane) * 2] = sm;
                    red[((size_t)wave * NCO + nb * 16 + lane) * 2 + 1] = sq;
                }
            } else {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    double sm = 0.0, sq = 0.0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double v = (double)fmaxf(acc[mb][nb][r], 0.f);
                        sm += v; sq += v * v;
                    }
                    sm += __shfl_xor(sm, 16, 64); sq += __shfl_xor(sq, 16, 64);      // lanes {0..31}: sample 2mb, {32..63}: 2mb+1
                    if ((lane & 16) == 0) {
                        const int slot = mb * 2 + (lane >> 5);
                        red[(((size_t)wave * SLOTS + slot) * NCO + nb * 16 + (lane & 15)) * 2] = sm;
                        red[(((size_t)wave * SLOTS + slot) * NCO + nb * 16 + (lane & 15)) * 2 + 1] = sq;
                    }
                }
            }
        }
        __syncthreads();
        const int tile = SPW == 1 ? (int)(lblock % ((edge / TZ) * (edge / TY) * (edge / TX))) : 0;
        for (int idx = tid; idx < SPW * NCO; idx += NT) {
            const int sidx = idx / NCO, col = idx % NCO;
            const int co = cob + col, nn = n0 + sidx;
            if (co < a.cout && nn < a.n) {
                double sm = 0.0, sq = 0.0;
                if (SLOTS == 1) {
                    constexpr int WPSMP = VOL / WV;             // waves per sample
#pragma unroll
                    for (int w = 0; w < WPSMP; ++w) {
                        sm += red[((size_t)(sidx * WPSMP + w) * NCO + col) * 2];
                        sq += red[((size_t)(sidx * WPSMP + w) * NCO + col) * 2 + 1];
                    }
                } else {
                    const int w = (sidx * VOL) / WV, slot = ((sidx * VOL) % WV) / VOL;
                    sm = red[(((size_t)w * SLOTS + slot) * NCO + col) * 2];
                    sq = red[(((size_t)w * SLOTS + slot) * NCO + col) * 2 + 1];
                }
                a.stats[((size_t)nn * a.cout + co) * a.stats_tiles + tile] = make_double2(sm, sq);
            }
        }
    }

    // ---- optional: fused MaxPool3d(2) of the ReLU'd box and the pooled tensor's GroupNorm statistics (8^3 boxes).  A lane's
    // accumulators hold the z pair (mb, mb+1) and the x pairs (r) of its pooling cells; the y pair sits in lane ^ 32.
    if constexpr (BoxOrder<TZ, TY, TX, NW, MB>::POOLABLE) {
        if (a.pool_mode) {
            const int hedge = edge >> 1, kq = lane >> 4;
            const size_t pvol = (size_t)hedge * hedge * hedge;
            double* red = reinterpret_cast<double*>(smem);      // [NW][NCO][2]
            if (a.stats) __syncthreads();                       // the statistics block above may still be reading `red`
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int co = cob + nb * 16 + (lane & 15);
                double sm = 0.0, sq = 0.0;
#pragma unroll
                for (int yh = 0; yh < 2; ++yh) {
                    const f32x4 u = acc[2 * yh][nb], v = acc[2 * yh + 1][nb];
                    float p0 = fmaxf(fmaxf(u[0], u[1]), fmaxf(v[0], v[1]));
                    float p1 = fmaxf(fmaxf(u[2], u[3]), fmaxf(v[2], v[3]));
                    p0 = fmaxf(p0, __shfl_xor(p0, 32, 64));
                    p1 = fmaxf(p1, __shfl_xor(p1, 32, 64));
                    p0 = fmaxf(p0, 0.f);                        // max and ReLU commute
                    p1 = fmaxf(p1, 0.f);
                    if (kq < 2) {
                        sm += (double)p0 + (double)p1;
                        sq += (double)p0 * (double)p0 + (double)p1 * (double)p1;
                        if (co < a.cout) {
                            const int pz = (z0 >> 1) + (wave >> 1), py = (y0 >> 1) + 2 * (wave & 1) + yh, px = (x0 >> 1) + 2 * kq;
                            *reinterpret_cast<float2*>(a.pool_out + ((size_t)n0 * a.cout + co) * pvol + ((size_t)pz * hedge + py) * hedge + px) =
                                make_float2(p0, p1);
                        }
                    }
                }
                if (a.pool_stats) {
                    sm += __shfl_xor(sm, 16, 64); sq += __shfl_xor(sq, 16, 64);       // the two x halves (lane groups 0 and 1)
                    if (lane < 16) {
                        red[((size_t)wave * NCO + nb * 16 + lane) * 2] = sm;
                        red[((size_t)wave * NCO + nb * 16 + lane) * 2 + 1] = sq;
                    }
                }
            }
            if (a.pool_stats) {
                __syncthreads();
                const int tile = (int)(lblock % ((edge / TZ) * (edge / TY) * (edge / TX)));
                for (int col = tid; col < NCO; col += NT) {
                    const int co = cob + col;
                    if (co < a.cout) {
                        double sm = 0.0, sq = 0.0;
#pragma unroll
                        for (int w = 0; w < NW; ++w) {
                            sm += red[((size_t)w * NCO + col) * 2];
                            sq += red[((size_t)w * NCO + col) * 2 + 1];
                        }
                        a.pool_stats[((size_t)n0 * a.cout + co) * a.stats_tiles + tile] = make_double2(sm, sq);
                    }
                }
            }
        }
    }
}
