// 3x3x3 conv + GroupNorm apply + ReLU on whole 8^3 volumes, position-major (see conv3d_small.hip for the idea): an MFMA
// m-block is ONE VOXEL POSITION of 16 DIFFERENT SAMPLES, so whether a tap reads zero padding is a compile-time property of
// the m-block and every padding tap is left out (23 % of the taps of an 8^3 volume; the box-tiled kernels can only leave
// out the z-border ones, 8 %).  Bit-identical to the box-tiled kernels: the MFMAs left out would have added exact zeros.
//
// A workgroup of 8 waves owns ONE z SLICE (64 positions) of 16 samples: wave w owns row y = w, m-block mb is x = mb.  The
// taps that exist depend on the class of z (first / interior / last slice: uniform over the workgroup) and of y (per
// wave): 9 variants of the whole K loop + epilogue, a wave picks one.  Staged per 4-channel chunk: the three input slices
// z-1, z, z+1 as [slice][position][4 ch][16 samples] (lane -> 16*k + sample: conflict-free operand reads, taps are
// immediate offsets).  Weight slabs by LDS-DMA, double buffered, as conv3d_mfma.hip.
//
// Decoder form (template flag UP, rf_conv3d_up_k3_gn_relu): phase B convolves the c1 low-res (4^3) channels with the
// parity-split pre-summed taps of conv3d_up.hip; z parity and lattice plane are uniform over the workgroup, so it stages
// two low-res planes and the 4 parities (py,px) of its z parity.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((address_space(1))) const void* rf_gptr;
typedef __attribute__((address_space(3))) void* rf_lptr;

struct Pm8Args {
    const float* src;      // [n][cin][8^3] (skip source in the decoder form; cin may be 0 there)
    const float* scale;    // [n][cin + c1]
    const float* shift;
    const float* wp;       // [27][cin4][cout16]
    float* out;
    int cin, n, cout, cin4, cout16;
    double2* stats;        // optional [n][cout][8] (one tile per z slice)
    const float* src1;     // decoder form: [n][c1][4^3]
    const float* wp1;      // [c1_8/8][8 parities][8 taps][8 ch][cout16]
    int c1, c1_8;
};

template <int NB, bool UP>
struct Pm8Tile {
    static constexpr int NT = 512, MB = 8;
    static constexpr int NCO = NB * 16;
    static constexpr int XS = 3 * 64 * 64;                    // [3 slices][64 positions][4 ch][16 samples]
    static constexpr int WSLAB = 27 * 4 * NCO;
    static constexpr int WSLAB_PAD = (WSLAB + 255) / 256 * 256;
    static constexpr int MAIN = XS + 2 * WSLAB_PAD;
    static constexpr int XLOW = 2 * 2 * 16 * 64;              // decoder form: two buffers of [2 planes][16 positions][4 ch][16 samples]
    static constexpr int BSLAB = 4 * 8 * 4 * NCO;             // [4 parities (py,px)][8 taps][4 ch][NCO], double buffered
    static constexpr int MAINB = UP ? XLOW + 2 * BSLAB : 0;
    static constexpr int EROW = 65;                           // epilogue tile [16 cout][16 samples][64 + 1]
    static constexpr int EPI = 16 * 16 * EROW;
    static constexpr int MAXAB = MAIN > MAINB ? MAIN : MAINB;
    static constexpr size_t LDS_BYTES = (size_t)(MAXAB > EPI ? MAXAB : EPI) * sizeof(float);
    static_assert(LDS_BYTES <= 81920, "two workgroups per CU");
};

template <int NB, bool UP>
__global__ __launch_bounds__(512, 4) void k_conv3_pm8(Pm8Args a) {
    using T = Pm8Tile<NB, UP>;
    constexpr int NCO = T::NCO, NT = T::NT, MB = T::MB, EROW = T::EROW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* wsb = smem + T::XS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // = y of this wave's row
    // the 8 z slices of a sample group re-read each other's input slices: keep them on one XCD (one L2).  Workgroups go to the
    // XCDs round-robin by linear id and gridDim.x is a multiple of 8, so XCD k must walk a contiguous range of (group, z) ids
    const unsigned lb = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int z = lb & 7, n0 = (lb >> 3) * 16, cob = blockIdx.y * NCO;
    const int kq = lane >> 4, li = lane & 15;
    const int ctot = a.cin + (UP ? a.c1 : 0);
    constexpr bool PFX = true;                                       // prefetch the next chunk's rows through registers under the MFMA loop

    constexpr int ROT = NCO >= 32 ? 16 : 0;
    int boff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) boff[nb] = kq * NCO + ((nb * 16 + li + ROT * (kq & 1)) % NCO);
    // operand base: middle slice, row y, x = 0; the tap / x of the m-block are immediates
    const int abase = (64 + wave * 8) * 64 + kq * 16 + li;

    constexpr int WF4 = T::WSLAB / 4, NPIECE = (WF4 + 63) / 64;
    auto dma_weights = [&](int cbase, int buf, int lane_) {
        float* dst = wsb + buf * T::WSLAB_PAD;
#pragma unroll
        for (int i = 0; i < (NPIECE + 7) / 8; ++i) {
            const int q = wave + i * 8;
            if (q < NPIECE) {
                int idx = q * 64 + lane_;
                if (idx >= WF4) idx = WF4 - 1;
                const int r = idx / (NCO / 4), slot = (idx % (NCO / 4)) * 4;
                const int col = (slot + NCO - ROT * (r & 1)) % NCO;
                int co = cob + col;
                if (co >= a.cout16) co = col % a.cout16;
                int ci = cbase + r % 4;
                if (ci >= a.cin4) ci = a.cin4 - 1;
                const float* src = a.wp + ((size_t)(r / 4) * a.cin4 + ci) * a.cout16 + co;
                __builtin_amdgcn_global_load_lds((rf_gptr)src, (rf_lptr)(dst + q * 256), 16, 0, 0);
            }
        }
    };

    // ---- input chunk: item = (slice, run of 8 positions, sample, channel): thread -> (sample, k) fastest, run = tid >> 6, 3 slices
    float xraw[3][8];
    float xsc = 0.f, xsh = 0.f;
    const int it_s = tid & 15, it_k = (tid >> 4) & 3, it_run = tid >> 6;
    auto issue_rows = [&](int cbase) {
        const int nn = n0 + it_s, ci = cbase + it_k;
        if (nn < a.n && ci < a.cin) {
            xsc = a.scale[(size_t)nn * ctot + ci];
            xsh = a.shift[(size_t)nn * ctot + ci];
            const float* base = a.src + ((size_t)nn * a.cin + ci) * 512 + it_run * 8;
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) {
                const int zg = z - 1 + sl;
                if ((unsigned)zg < 8u) {                             // slices outside the volume are never read (taps left out)
                    const float4* row = reinterpret_cast<const float4*>(base + zg * 64);
                    const float4 t0 = row[0], t1 = row[1];
                    xraw[sl][0] = t0.x; xraw[sl][1] = t0.y; xraw[sl][2] = t0.z; xraw[sl][3] = t0.w;
                    xraw[sl][4] = t1.x; xraw[sl][5] = t1.y; xraw[sl][6] = t1.z; xraw[sl][7] = t1.w;
                }
            }
        }
    };
    auto commit_rows = [&](int cbase) {
        const int nn = n0 + it_s, ci = cbase + it_k;
        const bool ok = nn < a.n && ci < a.cin;
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            float* dst = xs + (sl * 64 + it_run * 8) * 64 + it_k * 16 + it_s;
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[j * 64] = ok ? xraw[sl][j] * xsc + xsh : 0.f;
        }
    };

    // Everything below is instantiated per variant (ZC / YC: 0 = first, 1 = interior, 2 = last slice / row)
    auto run = [&](auto zc_c, auto yc_c) {
        constexpr int ZC = decltype(zc_c)::value, YC = decltype(yc_c)::value;
        f32x4 acc[MB][NB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

        if (a.cin > 0) {
            dma_weights(0, 0, lane);
            issue_rows(0);
            commit_rows(0);
        }
        __syncthreads();

        int buf = 0;
        for (int cbase = 0; cbase < a.cin; cbase += 4) {
            const bool more = cbase + 4 < a.cin;
            if (more) {
                if (PFX) issue_rows(cbase + 4);
                int lane_o = lane;
                asm volatile("" : "+v"(lane_o));
                dma_weights(cbase + 4, buf ^ 1, lane_o);
            }
            const float* ws = wsb + buf * T::WSLAB_PAD;
            // operands of tap t+1 are read while tap t multiplies (explicit double buffering; validity is compile-time)
            float avs[2][MB], bvs[2][NB];
            auto tap_live = [&](int t) -> bool {
                const int dz = t / 9, dy = (t / 3) % 3;
                return !((ZC == 0 && dz == 0) || (ZC == 2 && dz == 2) || (YC == 0 && dy == 0) || (YC == 2 && dy == 2));
            };
            auto tap_load = [&](int t, float (&av)[MB], float (&bv)[NB]) {
                const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bv[nb] = ws[boff[nb] + t * 4 * NCO];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int ux = mb + dx - 1;
                    if (ux >= 0 && ux <= 7) av[mb] = xs[abase + ((dz - 1) * 64 + (dy - 1) * 8 + ux) * 64];
                }
            };
            int slot = 0;
            bool primed = false;
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                if (!tap_live(t)) continue;                                          // compile-time: padding in z / y
                if (!primed) { tap_load(t, avs[0], bvs[0]); primed = true; }
                asm volatile("" ::: "memory");                                       // keep later taps' operand reads out of this one
                int tn = t + 1;
                while (tn < 27 && !tap_live(tn)) ++tn;
                if (tn < 27) tap_load(tn, avs[slot ^ 1], bvs[slot ^ 1]);
                const int dx = t % 3;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int ux = mb + dx - 1;
                    if (ux < 0 || ux > 7) continue;                                  // ... in x
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(avs[slot][mb], bvs[slot][nb], acc[mb][nb], 0, 0, 0);
                }
                slot ^= 1;
            }
            __syncthreads();
            if (more) {
                if (!PFX) issue_rows(cbase + 4);
                commit_rows(cbase + 4);
            }
            __syncthreads();
            buf ^= 1;
        }

        // ---- decoder form, phase B: output (z,y,x) has parity p = (z&1, y&1, x&1) and lattice cell (Z,Y,X) = (z,y,x) >> 1; its
        // low-res tap (tz,ty,tx) reads low-res voxel (Z + tz - 1 + pz, ...) with the pre-summed weight W'[parity][tap] -- or
        // zero padding, which is not issued.  Staged planes: index tz <-> low-res z = Z - 1 + pz + tz.
        if constexpr (UP) {
            float* xlow = smem;                                      // [2 buffers][2 planes][16 positions][4 ch][16 samples]
            float* bsl = smem + T::XLOW;                             // [2 buffers][((py*2+px)*8 + tap)*4 + k][NCO]
            constexpr int BROT = NCO == 32 ? 16 : 0;
            constexpr int BF4 = T::BSLAB / 4, BPIECE = BF4 / 64;
            const int pz = z & 1, Zl = z >> 1;
            auto dma_b = [&](int c4, int bufb, int lane_) {          // c4: 4-channel chunk of c1
                float* dst = bsl + bufb * T::BSLAB;
#pragma unroll
                for (int i = 0; i < (BPIECE + 7) / 8; ++i) {
                    const int q = wave + i * 8;
                    if (q < BPIECE) {
                        const int idx = q * 64 + lane_;
                        const int r = idx / (NCO / 4), slot = (idx % (NCO / 4)) * 4;
                        const int col = (slot + NCO - BROT * ((r >> 1) & 1)) % NCO;
                        int co = cob + col;
                        if (co >= a.cout16) co = col % a.cout16;
                        const int k = r & 3, pt = (r >> 2) + pz * 32;   // pt = parity*8 + tap, parity = pz*4 + py*2 + px
                        const float* src = a.wp1 + ((((size_t)(c4 >> 1) * 64 + pt) * 8) + (c4 & 1) * 4 + k) * a.cout16 + co;
                        __builtin_amdgcn_global_load_lds((rf_gptr)src, (rf_lptr)(dst + q * 256), 16, 0, 0);
                    }
                }
            };
            // low-res chunk: 2 planes x 16 positions x 4 ch x 16 samples: item = (plane, run of 8 positions, sample, k): 256 items
            float lraw[8];
            float lsc = 0.f, lsh = 0.f;
            const int l_run = (tid >> 6) & 1, l_pl = (tid >> 7) & 1;
            auto issue_low = [&](int c4) {
                const int nn = n0 + it_s, ci = c4 * 4 + it_k, zl = Zl - 1 + pz + l_pl;
                if (tid < 256 && nn < a.n && ci < a.c1 && (unsigned)zl < 4u) {
                    lsc = a.scale[(size_t)nn * ctot + a.cin + ci];
                    lsh = a.shift[(size_t)nn * ctot + a.cin + ci];
                    const float4* row = reinterpret_cast<const float4*>(a.src1 + ((size_t)nn * a.c1 + ci) * 64 + zl * 16 + l_run * 8);
                    const float4 t0 = row[0], t1 = row[1];
                    lraw[0] = t0.x; lraw[1] = t0.y; lraw[2] = t0.z; lraw[3] = t0.w;
                    lraw[4] = t1.x; lraw[5] = t1.y; lraw[6] = t1.z; lraw[7] = t1.w;
                }
            };
            auto commit_low = [&](int c4, int bufb) {
                if (tid < 256) {
                    const int nn = n0 + it_s, ci = c4 * 4 + it_k;
                    const bool ok = nn < a.n && ci < a.c1;
                    float* dst = xlow + bufb * 2048 + (l_pl * 16 + l_run * 8) * 64 + it_k * 16 + it_s;
#pragma unroll
                    for (int j = 0; j < 8; ++j) dst[j * 64] = ok ? lraw[j] * lsc + lsh : 0.f;
                }
            };
            const int nchunk = a.c1_8 >> 2;
            int bboff[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bboff[nb] = kq * NCO + ((nb * 16 + li + BROT * ((kq >> 1) & 1)) % NCO);
            const int py = wave & 1, Yl = wave >> 1;                 // y parity / lattice row of this wave
            dma_b(0, 0, lane);
            issue_low(0);
            commit_low(0, 0);
            __syncthreads();
            int bufb = 0;
            for (int c4 = 0; c4 < nchunk; ++c4) {
                const bool more = c4 + 1 < nchunk;
                if (more) {
                    issue_low(c4 + 1);
                    int lane_o = lane;
                    asm volatile("" : "+v"(lane_o));
                    dma_b(c4 + 1, bufb ^ 1, lane_o);
                }
                // operand base: plane tz, low-res row Yl - 1 + py (+ ty), x = 0
                const float* xl = xlow + bufb * 2048 + ((Yl - 1 + py) * 4) * 64 + kq * 16 + li;
                const float* wb = bsl + bufb * T::BSLAB + py * (2 * 8 * 4 * NCO);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int tz = t >> 2, ty = (t >> 1) & 1, tx = t & 1;
                    // z: first slice (Z=0,pz=0): tz = 0 is padding; last slice (Z=3,pz=1): tz = 1 is padding
                    if ((ZC == 0 && tz == 0) || (ZC == 2 && tz == 1)) continue;
                    // y: first row (Y=0,py=0): ty = 0 is padding; last row (Y=3,py=1): ty = 1 is padding
                    if ((YC == 0 && ty == 0) || (YC == 2 && ty == 1)) continue;
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        const int px = mb & 1, ux = (mb >> 1) + tx - 1 + px;
                        if (ux < 0 || ux > 3) continue;                                  // compile-time: padding tap in x
                        const float av = xl[(tz * 16 + ty * 4 + ux) * 64];
                        const int brow = (px * 8 + t) * 4;
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

        // ---- epilogue: ReLU'd accumulators -> LDS [16 cout][sample][64 + 1] -> contiguous float4 rows; statistics
        float* eb = smem;
        double* red = reinterpret_cast<double*>(smem);               // [8 waves][16 samples][16 cout][2], after the stores
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    eb[(li * 16 + kq * 4 + r) * EROW + wave * 8 + mb] = fmaxf(acc[mb][nb][r], 0.f);
            }
            __syncthreads();
            for (int q = tid; q < 16 * 16 * 16; q += NT) {
                const int p4 = q & 15, smp = (q >> 4) & 15, col = q >> 8;
                const int co = cob + nb * 16 + col, nn = n0 + smp;
                if (co < a.cout && nn < a.n) {
                    const float* e = eb + (col * 16 + smp) * EROW + p4 * 4;
                    *reinterpret_cast<float4*>(a.out + ((size_t)nn * a.cout + co) * 512 + z * 64 + p4 * 4) = make_float4(e[0], e[1], e[2], e[3]);
                }
            }
            __syncthreads();
            if (a.stats) {
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
                        a.stats[((size_t)nn * a.cout + co) * 8 + z] = make_double2(s0, s1);
                    }
                }
                __syncthreads();
            }
        }
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    const int zc = z == 0 ? 0 : (z == 7 ? 2 : 1), yc = wave == 0 ? 0 : (wave == 7 ? 2 : 1);
    if (zc == 0) {
        if (yc == 0) run(I0{}, I0{}); else if (yc == 1) run(I0{}, I1{}); else run(I0{}, I2{});
    } else if (zc == 1) {
        if (yc == 0) run(I1{}, I0{}); else if (yc == 1) run(I1{}, I1{}); else run(I1{}, I2{});
    } else {
        if (yc == 0) run(I2{}, I0{}); else if (yc == 1) run(I2{}, I1{}); else run(I2{}, I2{});
    }
}

template <int NB, bool UP>
static int launch_pm8(const Pm8Args& a, hipStream_t stream) {
    using T = Pm8Tile<NB, UP>;
    auto kern = k_conv3_pm8<NB, UP>;
    static bool attr_set = false;
    if (!attr_set) {
        if (T::LDS_BYTES > 65536) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES);
            if (e != hipSuccess) { rf_set_error("rf_conv3d_k3_gn_relu(pm8): cannot raise LDS limit: %s", hipGetErrorString(e)); return RF_E_LAUNCH; }
        }
        attr_set = true;
    }
    const unsigned gx = (unsigned)((a.n + 15) / 16) * 8, gy = (unsigned)((a.cout16 + T::NCO - 1) / T::NCO);
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(T::NT), T::LDS_BYTES, stream, a);
    RF_CHECK_LAUNCH("rf_conv3d_k3_gn_relu(pm8)");
    return RF_OK;
}

// whole 8^3 volumes, enough samples to fill the chip.  RFUSE_CONV_PM8: 0 = off (default), 1 = decoder form, 2 = plain convs too.
// Measured (B = 32, 8192 patches): staging three slices per chunk and per cout block costs what the skipped taps save --
// plain convs 16->16 496 vs 466 us box-tiled, 16->32 902 vs 887, 56->16 1561 vs 1533; decoder form 32+64->56: 5.04 vs
// 5.24 ms (-4 %), but with 5.0 GB instead of 1.0 GB fetched per launch (the input is re-read per slice and per cout block)
// and the MFMA pipe 68 % instead of 81 % busy.  Kept as an option; the box-tiled kernels stay the default at 8^3.
static int pm8_knob() {
    static const int knob = getenv("RFUSE_CONV_PM8") ? atoi(getenv("RFUSE_CONV_PM8")) : 0;
    return knob;
}
bool rf_conv3_pm8_takes(int c0, int c1, int n, int edge, int cout) {
    if (edge != 8 || n < 512) return false;
    if (c1 > 0) return pm8_knob() >= 1 && c0 >= 0;
    return pm8_knob() >= 2 && c0 > 1;
}
int rf_conv3_pm8_stats_tiles() { return 8; }

int rf_conv3_pm8_launch(const float* src0, int c0, const float* src1, int c1, int n, const float* scale, const float* shift,
                        const float* w_packed, int cout, float* out, double* stats, void* stream) {
    Pm8Args a;
    a.src = src0; a.scale = scale; a.shift = shift; a.wp = w_packed; a.out = out;
    a.cin = c0; a.n = n; a.cout = cout; a.cin4 = rf_round_up(c0, 4); a.cout16 = rf_round_up(cout, 16);
    a.stats = reinterpret_cast<double2*>(stats);
    a.src1 = src1; a.c1 = c1; a.c1_8 = rf_round_up(c1, 8);
    a.wp1 = c1 > 0 ? w_packed + (size_t)27 * a.cin4 * a.cout16 : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (c1 > 0) return a.cout16 <= 16 ? launch_pm8<1, true>(a, s) : launch_pm8<2, true>(a, s);
    return a.cout16 <= 16 ? launch_pm8<1, false>(a, s) : launch_pm8<2, false>(a, s);
}
