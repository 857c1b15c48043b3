// rf_conv3d_k3_gn_relu: 3x3x3 convolution (pad 1, no bias) as an fp32-MFMA implicit GEMM for gfx950 with the
// GroupNorm apply, the decoder's nearest-upsample + concat read and the ReLU fused in.
// Reference arithmetic: model/unet.py:19-76 (SingleConv 'gcr'), :297-308 and :354-360 (upsample + concat).
//
// GEMM view: M = output voxels, N = cout, K = cin*27.  A workgroup of NW waves owns P = NW*MB*16 output voxels --
// a box of one sample, or several whole small volumes -- and up to NB*16 output channels.  K is walked in chunks of
// 4 input channels (= the k of v_mfma_f32_16x16x4_f32).  Per chunk:
//
//   weights  [27 taps][4 ch][NB*16 cout] slab of the packed weight image  --LDS-DMA (global_load_lds, 16 B/lane)-->
//            LDS, DOUBLE buffered: the slab of chunk c+1 streams in while chunk c is being multiplied; no VGPRs.
//            The image is stored with odd k-rows rotated by 16 floats (done on the SOURCE address, the DMA destination
//            is lane-linear) so the k / k+1 halves of a B read hit different banks.
//   input    halo box [4 ch][TZ+2][TY+2][TX+2] of the chunk: one thread per halo row; the row's global loads for
//            chunk c+1 are issued BEFORE the MFMA loop of chunk c into registers and committed to LDS after it with
//            GroupNorm (x*scale+shift) applied, zeros outside the volume, upsample/concat resolved.
//   compute  27 taps: A operands are im2col reads straight out of the halo box (lane -> voxel lane&15, channel lane>>4;
//            the tap offset is a compile-time LDS immediate), B operands rows of the slab; MB x NB accumulator tiles
//            per wave.  fp32-input MFMA == fp32 FMA chain in k order, so this is exact fp32 arithmetic.
//
// Tile choice (measured): 8 waves x MB=4 (512 voxels) keeps accumulators at 64 VGPRs so two workgroups =
// 4 waves/SIMD stay resident -- the MFMA loop of the earlier 4-wave x MB=8 tile (128 accumulator VGPRs, 2 waves/SIMD)
// was only 78 % busy on its own.  Small problems take 4 waves x MB=2 (128 voxels) to fill the chip.
#include "common.h"
#include "conv_box.h"
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((address_space(1))) const void* rf_gptr;
typedef __attribute__((address_space(3))) void* rf_lptr;




template <int TZ, int TY, int TX, int SPW, int NW, int MB, int NB, int CC_>
struct ConvTile {
    static constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
    static constexpr int CH = HZ * HY * HX;            // halo floats per channel
    static constexpr int CC = CC_;                      // input channels per K chunk: 4 (one MFMA k-step per tap) or 8 (two)
    static_assert(CC_ == 4 || CC_ == 8, "K chunk is one or two MFMA k-steps deep");
    static constexpr int XS = SPW * CC * CH;            // floats of the staged input box
    static constexpr int XS_PAD = (XS + 3) / 4 * 4;
    static constexpr int NCO = NB * 16;                 // couts per workgroup
    static constexpr int WSLAB = 27 * CC * NCO;         // floats of one weight slab (unpadded: the DMA image is linear)
    static constexpr int WSLAB_PAD = (WSLAB + 255) / 256 * 256;   // whole 1-KiB DMA pieces
    static constexpr int NT = NW * 64;
    static constexpr int P = SPW * TZ * TY * TX;
    static constexpr size_t LDS_BYTES = (size_t)(XS_PAD + 2 * WSLAB_PAD) * sizeof(float);
    static_assert(P == NW * MB * 16, "NW waves x MB x 16 voxels must cover the tile");
    static_assert(SPW == 1 || TX < 8, "multi-sample tiles are for whole small volumes");
};

// PFX: prefetch the next chunk's input rows through registers under the MFMA loop (costs RPT*(TX+4) VGPRs)
template <int TZ, int TY, int TX, int SPW, int NW, int MB, int NB, int WPS, bool PFX, int CCT>
__global__ __launch_bounds__(NW * 64, WPS) void k_conv3_mfma(ConvArgs a) {
    using T = ConvTile<TZ, TY, TX, SPW, NW, MB, NB, CCT>;
    constexpr int HY = T::HY, HX = T::HX, CH = T::CH, CC = T::CC, NCO = T::NCO, NT = T::NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* wsb = smem + T::XS_PAD;                       // two slabs of WSLAB_PAD floats

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int edge = a.edge, cin = a.c0 + a.c1;
    const int half = edge >> 1;

    // ---- which voxels does this workgroup own?
    int n0, z0 = 0, y0 = 0, x0 = 0;
    const unsigned lblock = (SPW == 1 && gridDim.y == 1) ? rf_xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
    if (SPW == 1) {
        const int tx = edge / TX, ty = edge / TY, tz = edge / TZ;
        int t = (int)lblock;
        x0 = (t % tx) * TX; t /= tx;
        y0 = (t % ty) * TY; t /= ty;
        z0 = (t % tz) * TZ; t /= tz;
        n0 = t;
    } else {
        n0 = blockIdx.x * SPW;
    }
    const int cob = blockIdx.y * NCO;

    // ---- per-lane LDS read offsets
    int aoff[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        int s, z, y, x;
        BoxOrder<TZ, TY, TX, NW, MB>::voxel(wave, mb, lane & 15, s, z, y, x);
        aoff[mb] = s * (CC * CH) + (z * HY + y) * HX + x + (lane >> 4) * CH;
    }
    // which of this wave's m-blocks lie on the first / last z slice of the volume (wave-uniform bit masks over mb).  All 16
    // voxels of an m-block share z in the pooling order of the 8^3 box and when a z slice of the tile is a multiple of 16
    // voxels.  ZSKIP_LO / ZSKIP_HI are the only non-empty masks the tile shape can produce (see the MFMA loop).
    constexpr bool ZUNIFORM = BoxOrder<TZ, TY, TX, NW, MB>::POOLABLE || (TX * TY) % 16 == 0;
    constexpr unsigned ZSKIP_LO = !ZUNIFORM ? 0u : BoxOrder<TZ, TY, TX, NW, MB>::POOLABLE ? 0x5u : (TX * TY == 16 ? 0x1u : (1u << MB) - 1u);
    constexpr unsigned ZSKIP_HI = !ZUNIFORM ? 0u : BoxOrder<TZ, TY, TX, NW, MB>::POOLABLE ? 0xAu : (TX * TY == 16 ? (1u << (MB - 1)) : (1u << MB) - 1u);
    unsigned zlo_mask = 0u, zhi_mask = 0u;
    if (ZUNIFORM) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            int s, z, y, x;
            BoxOrder<TZ, TY, TX, NW, MB>::voxel(wave, mb, 0, s, z, y, x);
            if (z0 + z == 0) zlo_mask |= 1u << mb;
            if (z0 + z == edge - 1) zhi_mask |= 1u << mb;
        }
    }
    // slab row r = tap*4 + k holds cout column col at float (col + ROT*(r&1)) % NCO: rows k and k+1 on different banks
    constexpr int ROT = NCO >= 32 ? 16 : 0;
    int boff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
        boff[nb] = (lane >> 4) * NCO + ((nb * 16 + (lane & 15) + ROT * ((lane >> 4) & 1)) % NCO);

    f32x4 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- weight slab of one chunk -> LDS by DMA: piece q (1 KiB = 64 lanes x 16 B = RPP slab rows [tap][channel] x NCO) is issued by
    // wave q % NW.  Everything that depends on the lane is computed ONCE, per chunk only a scalar base changes: VALU instructions
    // issue through the same port as the MFMAs, and the round-1 form (~45 VALU per piece -- a runtime modulo and 64-bit multiplies
    // among them -- recomputed to save registers) cost the 16-cout layers a quarter of their MFMA time.
    static_assert(CC == 4, "the lane-constant DMA addressing assumes 4-channel chunks (cin4 % CC == 0)");
    constexpr int LPR = NCO / 4, RPP = 64 / LPR;            // lanes per slab row; slab rows per piece (4, 8, 16)
    constexpr int NPIECE = (27 * CC + RPP - 1) / RPP;
    const int wr = lane / LPR, wslot = (lane % LPR) * 4;
    const int wcol = (wslot + NCO - ROT * (wr & 1)) % NCO;   // logical cout column stored at this slot (slab row parity = piece row parity)
    int wco = cob + wcol;
    if (wco >= a.cout16) wco = wcol % a.cout16;              // cout block wider than the packed image: any valid column (masked at store)
    const unsigned wlane = 4u * (unsigned)(((wr / CC) * a.cin4 + wr % CC) * a.cout16 + wco);     // bytes
    auto dma_weights = [&](int cbase, int buf) {
        float* dst = wsb + buf * T::WSLAB_PAD;
        unsigned off = wlane;                                // opaque: SGPR base + 32-bit lane offset addressing at the instruction
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int i = 0; i < (NPIECE + NW - 1) / NW; ++i) {
            const int q = wave + i * NW;                     // wave-uniform
            if (q < NPIECE) {
                const int tap0 = q * (RPP / CC);
                const char* src = reinterpret_cast<const char*>(a.wp + ((size_t)tap0 * a.cin4 + cbase) * a.cout16);
                if (RPP == CC || tap0 + wr / CC < 27)        // rows past tap 26 of the last piece are never read
                    __builtin_amdgcn_global_load_lds((rf_gptr)(src + off), (rf_lptr)(dst + q * 256), 16, 0, 0);
            }
        }
    };

    // ---- input halo rows: one thread per row (HX = TX + 2 floats), register-staged.  Row r -> (sample s, chunk channel c, hz, hy);
    // the per-row constants once, per chunk a scalar base (the skip-source rows; rows of an upsampled source keep the index math).
    constexpr int ROWS = SPW * CC * T::HZ * HY;
    constexpr int RPT = (ROWS + NT - 1) / NT;
    float xraw[RPT][TX + 2];       // [left halo | TX interior (or TX/2 low-res values) | right halo]
    float xce[RPT], xsc[RPT], xsh[RPT];
    const bool has_l = x0 > 0, has_r = x0 + TX < edge;
    unsigned rowoff[RPT], affoff[RPT];                     // bytes behind (sample n0, channel cbase) of src0 / of the affine table
    int rowbox[RPT], rowc[RPT];                            // LDS row; chunk channel (>= 1 << 20: row outside the volume -> zeros, < 0: no row)
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = tid + i * NT;
        const int hy = r % HY, hz = (r / HY) % T::HZ, c = (r / (HY * T::HZ)) % CC, s = r / (HY * T::HZ * CC);
        const int z = z0 + hz - 1, y = y0 + hy - 1;
        const bool in = r < ROWS && n0 + s < a.n && (unsigned)z < (unsigned)edge && (unsigned)y < (unsigned)edge;
        rowoff[i] = in ? 4u * (unsigned)((((s * a.c0 + c) * edge + z) * edge + y) * edge + x0) : 0u;
        affoff[i] = in ? 16u * (unsigned)(s * cin + c) : 0u;
        rowbox[i] = (s * CC + c) * CH + (hz * HY + hy) * HX;
        rowc[i] = r < ROWS ? (in ? c : (1 << 20)) : -1;
    }

    auto issue_rows = [&](int cbase) {
        const char* vol0 = reinterpret_cast<const char*>(a.src0 + ((size_t)n0 * a.c0 + cbase) * edge * edge * edge);
        const char* aff0 = reinterpret_cast<const char*>(a.affine + ((size_t)n0 * cin + cbase));
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            if (rowc[i] >= 0 && cbase + rowc[i] < cin) {
                unsigned ro = rowoff[i], ao = affoff[i];
                asm volatile("" : "+v"(ro), "+v"(ao));
                { const float4 af = *reinterpret_cast<const float4*>(aff0 + ao); xce[i] = af.x; xsc[i] = af.y; xsh[i] = af.z; }
                if (cbase + rowc[i] < a.c0) {
                    const float* row = reinterpret_cast<const float*>(vol0 + ro);
                    if (TX >= 4) {
#pragma unroll
                        for (int q = 0; q < TX / 4; ++q) {
                            const float4 t = reinterpret_cast<const float4*>(row)[q];
                            xraw[i][1 + 4 * q] = t.x; xraw[i][2 + 4 * q] = t.y; xraw[i][3 + 4 * q] = t.z; xraw[i][4 + 4 * q] = t.w;
                        }
                    } else {
                        const float2 t = *reinterpret_cast<const float2*>(row);
                        xraw[i][1] = t.x; xraw[i][2] = t.y;
                    }
                    if (has_l) xraw[i][0] = row[-1];
                    if (has_r) xraw[i][TX + 1] = row[TX];
                } else {
                    // nearest x2 upsample: voxel x reads low-res x>>1 -> TX/2 low-res values, expanded at commit time
                    const int r = tid + i * NT;
                    const int hy = r % HY, hz = (r / HY) % T::HZ, s = r / (HY * T::HZ * CC);
                    const int z = z0 + hz - 1, y = y0 + hy - 1;
                    const float* row = a.src1 + ((((size_t)(n0 + s) * a.c1 + (cbase + rowc[i] - a.c0)) * half + (z >> 1)) * half + (y >> 1)) * half + (x0 >> 1);
                    if (TX == 8) {
                        const float4 t = *reinterpret_cast<const float4*>(row);
                        xraw[i][1] = t.x; xraw[i][2] = t.y; xraw[i][3] = t.z; xraw[i][4] = t.w;
                    } else if (TX == 4) {
                        const float2 t = *reinterpret_cast<const float2*>(row);
                        xraw[i][1] = t.x; xraw[i][2] = t.y;
                    } else {
                        xraw[i][1] = row[0];
                    }
                    if (has_l) xraw[i][0] = row[-1];
                    if (has_r) xraw[i][TX + 1] = row[TX / 2];
                }
            }
        }
    };

    auto commit_rows = [&](int cbase) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            if (rowc[i] >= 0) {
                float v[TX + 2];
                if (cbase + rowc[i] < cin) {
                    const float ce = xce[i], sc = xsc[i], sh = xsh[i];
                    if (cbase + rowc[i] < a.c0) {
#pragma unroll
                        for (int j = 1; j <= TX; ++j) v[j] = fmaf(xraw[i][j] - ce, sc, sh);
                    } else {
#pragma unroll
                        for (int j = 0; j < TX; ++j) v[1 + j] = fmaf(xraw[i][1 + (j >> 1)] - ce, sc, sh);
                    }
                    v[0] = has_l ? fmaf(xraw[i][0] - ce, sc, sh) : 0.f;
                    v[TX + 1] = has_r ? fmaf(xraw[i][TX + 1] - ce, sc, sh) : 0.f;
                } else {
#pragma unroll
                    for (int j = 0; j < TX + 2; ++j) v[j] = 0.f;
                }
                float* dst = xs + rowbox[i];
#pragma unroll
                for (int j = 0; j < TX + 2; ++j) dst[j] = v[j];
            }
        }
    };

    // ---- prologue: chunk 0
    dma_weights(0, 0);
    issue_rows(0);
    commit_rows(0);
    __syncthreads();                                       // also drains the DMA (vmcnt(0) before the barrier)

    // K loop + epilogue, instantiated per z-border variant of this wave (see the MFMA loop); barriers match across variants
    auto run = [&](auto lo_c, auto hi_c) {
    constexpr unsigned LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
    int buf = 0;
    for (int cbase = 0; cbase < cin; cbase += CC) {
        const bool more = cbase + CC < cin;
        if (more) {
            if (PFX) issue_rows(cbase + CC);               // global loads fly under the MFMA loop ...
            dma_weights(cbase + CC, buf ^ 1);              // ... and so does the next weight slab
        }
        {
            const float* ws = wsb + buf * T::WSLAB_PAD;
            // The 27-tap MFMA loop, specialised at compile time on which m-blocks skip their dz = -1 (LO) / dz = +1 (HI) taps:
            // an m-block on the first / last z slice of the VOLUME reads nothing but zero padding through those taps, so the
            // MFMAs would add exact zeros -- same result bit for bit, 1/3 of that m-block's work saved.  LO / HI are
            // parameters of the enclosing `run` lambda: the whole K loop + epilogue is instantiated per variant, because a
            // run-time test (or a per-chunk variant switch) makes hipcc shuffle the accumulators through copies at every join.
            {
                // operand reads of tap t+1 are issued before the MFMAs of tap t (explicit double buffering: left to itself
                // hipcc emits read -> s_waitcnt lgkmcnt(0) -> MFMAs per tap pair and exposes the whole LDS latency)
                constexpr int KS = CC / 4;                         // MFMA k-steps per tap
                float av[2][MB], bv[2][NB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) av[0][mb] = xs[aoff[mb]];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bv[0][nb] = ws[boff[nb]];
#pragma unroll
                for (int st = 0; st < 27 * KS; ++st) {             // step = (tap, k-step)
                    const int cur = st & 1, nxt = cur ^ 1;
                    // keep the operand reads of later taps out of this step: left alone, hipcc pairs reads of neighbouring taps
                    // into ds_read2 far ahead of their use (long live ranges: +30..50 VGPRs, spills in the parity-split kernel)
                    __builtin_amdgcn_sched_barrier(0);
                    if (st + 1 < 27 * KS) {
                        const int t1 = (st + 1) / KS, k1 = (st + 1) % KS;
                        const int toff = ((t1 / 9) * HY + (t1 / 3) % 3) * HX + t1 % 3 + k1 * 4 * CH;
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) av[nxt][mb] = xs[aoff[mb] + toff];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) bv[nxt][nb] = ws[boff[nb] + (t1 * CC + k1 * 4) * NCO];
                    }
                    const int dz = (st / KS) / 9;
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        if ((dz == 0 && ((LO >> mb) & 1u)) || (dz == 2 && ((HI >> mb) & 1u))) continue;     // compile-time
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][mb], bv[cur][nb], acc[mb][nb], 0, 0, 0);
                    }
                    // the reads of step st+1 go out one behind every MFMA of step st (the wave's instruction stream stays MFMA-dense)
#pragma unroll
                    for (int i = 0; i < MB * NB; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
            }
        }
        __syncthreads();                                   // everyone is done reading xs / ws[buf]; loads + DMA have landed
        if (more) {
            if (!PFX) issue_rows(cbase + CC);
            commit_rows(cbase + CC);
        }
        __syncthreads();
        buf ^= 1;
    }

        conv_box_epilogue<TZ, TY, TX, SPW, NW, MB, NB, T::LDS_BYTES>(a, acc, smem, tid, lane, wave, n0, z0, y0, x0, cob, lblock);
    };
    using Zc = std::integral_constant<unsigned, 0u>;
    if (ZSKIP_LO != 0u && zlo_mask == ZSKIP_LO && zhi_mask == 0u) run(std::integral_constant<unsigned, ZSKIP_LO>{}, Zc{});
    else if (ZSKIP_HI != 0u && zhi_mask == ZSKIP_HI && zlo_mask == 0u) run(Zc{}, std::integral_constant<unsigned, ZSKIP_HI>{});
    else run(Zc{}, Zc{});
}

template <int TZ, int TY, int TX, int SPW, int NW, int MB, int NB, int WPS, int CCT = 4>
static int launch_conv3(const ConvArgs& a, hipStream_t stream) {
    using T = ConvTile<TZ, TY, TX, SPW, NW, MB, NB, CCT>;
    auto kern = k_conv3_mfma<TZ, TY, TX, SPW, NW, MB, NB, WPS, true, CCT>;
    if (T::LDS_BYTES > 65536) {
        static RfLdsOptIn opt_in;
        if (int rc = opt_in.ensure(reinterpret_cast<const void*>(kern), (int)T::LDS_BYTES, "rf_conv3d_k3_gn_relu")) return rc;
    }
    unsigned gx;
    if (SPW == 1) gx = (unsigned)a.n * (a.edge / TZ) * (a.edge / TY) * (a.edge / TX);
    else gx = (unsigned)((a.n + SPW - 1) / SPW);
    const unsigned gy = (unsigned)((a.cout16 + T::NCO - 1) / T::NCO);
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(T::NT), T::LDS_BYTES, stream, a);
    RF_CHECK_LAUNCH("rf_conv3d_k3_gn_relu");
    return RF_OK;
}

// big: 8 waves x MB 4 = 512 voxels (two workgroups = 4 waves/SIMD per CU); small: 4 waves x MB 2 = 128 voxels
enum { TILE_SMALL = 0, TILE_BIG = 1, TILE_MID = 2 };
template <int TZ, int TY, int TX, int SPW, int MODE>
static int dispatch_nb(const ConvArgs& a, hipStream_t stream) {
    if constexpr (MODE == TILE_MID) {                     // 8 waves x MB 2: 256 voxels = 4 / 32 whole 4^3 / 2^3 samples
        if (a.cout16 <= 16) return launch_conv3<TZ, TY, TX, SPW, 8, 2, 1, 4>(a, stream);
        if (a.cout16 <= 32) return launch_conv3<TZ, TY, TX, SPW, 8, 2, 2, 4>(a, stream);
        return launch_conv3<TZ, TY, TX, SPW, 8, 2, 4, 4>(a, stream);
    } else if constexpr (MODE == TILE_BIG) {
        // (16-cout layers: an 8-channel K chunk -- 216 instead of 108 MFMAs per wave between barriers -- measured 3-7 % slower than
        // the 4-channel chunk on every such layer and is not instantiated)
        if (a.cout16 <= 16) return launch_conv3<TZ, TY, TX, SPW, 8, 4, 1, 6>(a, stream);   // <= 85 VGPRs: three workgroups per CU
        if (a.cout16 <= 32) return launch_conv3<TZ, TY, TX, SPW, 8, 4, 2, 4>(a, stream);
        return launch_conv3<TZ, TY, TX, SPW, 8, 4, 4, 4>(a, stream);
    } else {
        if (a.cout16 <= 16) return launch_conv3<TZ, TY, TX, SPW, 4, 2, 1, 4>(a, stream);
        if (a.cout16 <= 32) return launch_conv3<TZ, TY, TX, SPW, 4, 2, 2, 4>(a, stream);
        return launch_conv3<TZ, TY, TX, SPW, 4, 2, 4, 4>(a, stream);
    }
}

// ---------------------------------------------------------------------------------------------- cin == 1 layers
// First conv of every U-Net (1 -> nf/2 channels, model/unet.py:128-132): K = 27 only, so the MFMA form would waste 3/4 of
// its k and half of its n.  It is HBM-bound (reads 4 B, writes 4*COUT B per voxel): plain VALU, one voxel per thread
// and all COUT channels in registers, halo tile and the [27][COUT] weights in LDS (weight reads are LDS broadcasts).
template <int TZ, int TY, int TX, int COUT>
__global__ __launch_bounds__(256) void k_conv3_cin1(ConvArgs a) {
    // one thread per (y, x) column of the tile, TZ voxels deep: each weight read (an LDS broadcast) feeds TZ voxels and
    // each input read feeds up to 3 of them, so LDS traffic per voxel is ~3x lower than a voxel-per-thread form
    constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
    static_assert(TY * TX == 256, "one thread per (y,x) column");
    __shared__ float xs[HZ * HY * HX];
    __shared__ __attribute__((aligned(16))) float wl[27 * 8];          // [tap][8] (COUT <= 8, zero padded)
    const int tid = threadIdx.x, edge = a.edge;
    const int tx = edge / TX, ty = edge / TY, tz = edge / TZ;
    const unsigned lblock = rf_xcd_contiguous(blockIdx.x, gridDim.x);
    int t = (int)lblock;
    const int x0 = (t % tx) * TX; t /= tx;
    const int y0 = (t % ty) * TY; t /= ty;
    const int z0 = (t % tz) * TZ; t /= tz;
    const int nn = t;
    for (int i = tid; i < 27 * 8; i += 256) wl[i] = (i % 8) < COUT ? a.wp[(size_t)(i / 8) * a.cin4 * a.cout16 + (i % 8)] : 0.f;   // packed [tap][ci=0][co]
    const float4 af = a.affine[nn];
    const float* src = a.src0 + (size_t)nn * edge * edge * edge;
    const int x = tid % TX, y = tid / TX;
    {
        // halo tile: a thread loads its own (y, x) column, HZ values, and the first 2 (TY + TX + 2) threads one ring element per plane -- no
        // div / mod chains (they were a quarter of the kernel's VALU instructions), all loads of a thread in flight at once (the rolled loop
        // waited for each before it asked for the next).  Out-of-volume slots read the tile's first voxel and are zeroed afterwards.
        constexpr int RING = 2 * (TY + TX + 2);
        static_assert(RING <= 256, "one ring element per thread and plane");
        int ry, rx;                                                  // ring element of this thread (tid < RING): halo-plane coordinates
        if (tid < HX) { ry = 0; rx = tid; }
        else if (tid < 2 * HX) { ry = HY - 1; rx = tid - HX; }
        else { const int r = tid - 2 * HX; ry = 1 + (r >> 1); rx = (r & 1) * (HX - 1); }
        const size_t first = ((size_t)z0 * edge + y0) * edge + x0;
        float raw[HZ], rraw[HZ];
        bool in[HZ], rin[HZ];
#pragma unroll
        for (int hz = 0; hz < HZ; ++hz) {
            const int z = z0 + hz - 1;
            const bool zin = (unsigned)z < (unsigned)edge;
            in[hz] = zin;
            raw[hz] = src[zin ? ((size_t)z * edge + (y0 + y)) * edge + (x0 + x) : first];
            const int gy = y0 + ry - 1, gx = x0 + rx - 1;
            rin[hz] = tid < RING && zin && (unsigned)gy < (unsigned)edge && (unsigned)gx < (unsigned)edge;
            rraw[hz] = src[rin[hz] ? ((size_t)z * edge + gy) * edge + gx : first];
        }
#pragma unroll
        for (int hz = 0; hz < HZ; ++hz) {
            xs[(hz * HY + y + 1) * HX + x + 1] = in[hz] ? fmaf(raw[hz] - af.x, af.y, af.z) : 0.f;
            if (tid < RING) xs[(hz * HY + ry) * HX + rx] = rin[hz] ? fmaf(rraw[hz] - af.x, af.y, af.z) : 0.f;
        }
    }
    __syncthreads();
    // cout PAIRS per accumulator register pair: v_pk_fma_f32 (two fp32 FMAs per lane and issue -- the 157 TFLOP/s vector peak is the packed
    // rate); per output the taps are still accumulated one by one in (dy, dx, dz) order, each with a single rounding
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int CP = (COUT + 1) / 2;
    f32x2 acc2[TZ][CP];
#pragma unroll
    for (int z = 0; z < TZ; ++z)
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) acc2[z][cp] = (f32x2){0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            float col[HZ];
#pragma unroll
            for (int hz = 0; hz < HZ; ++hz) col[hz] = xs[(hz * HY + (y + dy)) * HX + x + dx];
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) {
                const float4 w0 = *reinterpret_cast<const float4*>(wl + ((dz * 3 + dy) * 3 + dx) * 8);
                const float4 w1 = *reinterpret_cast<const float4*>(wl + ((dz * 3 + dy) * 3 + dx) * 8 + 4);
                const f32x2 wv[4] = {(f32x2){w0.x, w0.y}, (f32x2){w0.z, w0.w}, (f32x2){w1.x, w1.y}, (f32x2){w1.z, w1.w}};
#pragma unroll
                for (int z = 0; z < TZ; ++z) {
                    const f32x2 c2 = (f32x2){col[z + dz], col[z + dz]};
#pragma unroll
                    for (int cp = 0; cp < CP; ++cp) acc2[z][cp] = __builtin_elementwise_fma(c2, wv[cp], acc2[z][cp]);
                }
            }
        }
    float acc[TZ][COUT];
#pragma unroll
    for (int z = 0; z < TZ; ++z)
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[z][co] = acc2[z][co >> 1][co & 1];
    const size_t vol = (size_t)edge * edge * edge;
#pragma unroll
    for (int z = 0; z < TZ; ++z) {
        float* o = a.out + (size_t)nn * COUT * vol + ((size_t)(z0 + z) * edge + (y0 + y)) * edge + (x0 + x);
#pragma unroll
        for (int co = 0; co < COUT; ++co) o[co * vol] = fmaxf(acc[z][co], 0.f);
    }
    if (a.stats) {
        // per (wave, cout) sums of the ReLU'd outputs in float64.  The 16 values of a lane (sum and sum of squares of 8 couts) are reduced over
        // the wave by recursive halving: at step k a lane keeps half of its values and hands the other half to lane ^ (1 << k) -- 17 value
        // exchanges instead of 16 x 6 (the per-value butterflies were a quarter of the kernel's instructions); fixed order, no atomics
        __shared__ double red[4 * 16];
        const int lane = tid & 63, wave = tid >> 6;
        double v[16];
#pragma unroll
        for (int co = 0; co < 8; ++co) {
            double sm = 0.0, sq = 0.0;
            if (co < COUT) {
#pragma unroll
                for (int z = 0; z < TZ; ++z) {
                    const double t = (double)fmaxf(acc[z][co], 0.f);
                    sm += t; sq += t * t;
                }
            }
            v[2 * co] = sm; v[2 * co + 1] = sq;
        }
        auto halve = [&](auto kc) {
            constexpr int K = decltype(kc)::value, C = 8 >> K;
            const bool up = (lane >> K) & 1;
#pragma unroll
            for (int i = 0; i < C; ++i) {
                const double send = up ? v[i] : v[i + C], keep = up ? v[i + C] : v[i];
                v[i] = keep + __shfl_xor(send, 1 << K, 64);
            }
        };
        halve(std::integral_constant<int, 0>{});
        halve(std::integral_constant<int, 1>{});
        halve(std::integral_constant<int, 2>{});
        halve(std::integral_constant<int, 3>{});
        v[0] += __shfl_xor(v[0], 16, 64);
        v[0] += __shfl_xor(v[0], 32, 64);
        if (lane < 16) red[wave * 16 + ((lane & 1) * 8 + ((lane >> 1) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 3) & 1))] = v[0];
        __syncthreads();
        if (tid < COUT) {
            double sm = 0.0, sq = 0.0;
            for (int w = 0; w < 4; ++w) { sm += red[w * 16 + 2 * tid]; sq += red[w * 16 + 2 * tid + 1]; }
            const int tile = (int)(lblock % (tx * ty * tz));
            a.stats[((size_t)nn * COUT + tid) * a.stats_tiles + tile] = make_double2(sm, sq);
        }
    }
}

template <int COUT>
static int launch_cin1(const ConvArgs& a, hipStream_t stream) {      // edge >= 16
    const unsigned g = (unsigned)a.n * (a.edge / 4) * (a.edge / 16) * (a.edge / 16);
    hipLaunchKernelGGL((k_conv3_cin1<4, 16, 16, COUT>), dim3(g), dim3(256), 0, stream, a);
    RF_CHECK_LAUNCH("rf_conv3d_k3_gn_relu(cin1)");
    return RF_OK;
}

// which tiling will rf_conv3d_k3_gn_relu use, and how many stats tiles per sample does that give?
static bool conv_use_cin1(int c0, int c1, int edge, int cout) { return c0 == 1 && c1 == 0 && edge >= 16 && (cout == 8 || cout == 6); }
static bool conv_use_big(int n, int edge, int cout16) { return rf_conv_use_big(n, edge, cout16); }

extern "C" int rf_conv3d_stats_tiles(int c0, int c1, int n, int edge, int cout) {
    if (edge < 2) return 0;                                    // direct path: no fused statistics
    if (conv_use_cin1(c0, c1, edge, cout)) return (edge / 4) * (edge / 16) * (edge / 16);
    if (edge <= 4) return 1;                                   // whole-volume tiles (always the 128-voxel form)
    return conv_use_big(n, edge, rf_round_up(cout, 16)) ? (edge / 8) * (edge / 8) * (edge / 8) : (edge / 4) * (edge / 4) * (edge / 8);
}

bool rf_conv3_small_takes(int c0, int c1, int n, int edge, int cout);                       // conv3d_small.hip
int rf_conv3_small_launch(const float* src, int cin, int n, int edge, const float* gn_affine, const float* w_packed, int cout,
                          float* out, double* stats, void* stream, float* pool_out, double* pool_stats);

static int conv3d_impl(const float* src0, int c0, const float* src1, int c1, int n, int edge,
                       const float* gn_affine, const float* w_packed, int cout,
                       float* out, double* stats, void* stream, float* pool_out = nullptr, double* pool_stats = nullptr, int pool_mode = 0,
                       bool relu = true) {
    RF_REQUIRE(n > 0 && c0 >= 0 && c1 >= 0 && c0 + c1 > 0 && cout > 0, RF_E_INVALID, "rf_conv3d_k3_gn_relu: bad sizes");
    RF_REQUIRE(rf_is_pow2(edge) && edge <= 128, RF_E_INVALID, "rf_conv3d_k3_gn_relu: edge %d must be a power of two <= 128", edge);
    RF_REQUIRE((c0 == 0 || src0) && (c1 == 0 || src1) && gn_affine && w_packed && (out || pool_mode == 2), RF_E_INVALID,
               "rf_conv3d_k3_gn_relu: null pointer");
    RF_REQUIRE(edge >= 2, RF_E_UNSUPPORTED, "rf_conv3d_k3_gn_relu: 1^3 volumes take the direct path (rf_conv3d_k3_gn_relu_direct)");
    ConvArgs a;
    a.src0 = src0; a.src1 = src1; a.affine = reinterpret_cast<const float4*>(gn_affine); a.wp = w_packed; a.out = out;
    a.c0 = c0; a.c1 = c1; a.n = n; a.edge = edge; a.cout = cout;
    a.cin4 = rf_round_up(c0 + c1, 4); a.cout16 = rf_round_up(cout, 16);
    a.stats = reinterpret_cast<double2*>(stats);
    a.stats_tiles = (stats || pool_stats) ? rf_conv3d_stats_tiles(c0, c1, n, edge, cout) : 0;
    a.pool_out = pool_out; a.pool_stats = reinterpret_cast<double2*>(pool_stats); a.pool_mode = pool_mode;
    a.floor = relu ? 0.f : -INFINITY;
    hipStream_t s = (hipStream_t)stream;
    if (!relu) {                                          // only the box-tiled kernel carries the clamp switch
        RF_REQUIRE(!stats && !pool_mode, RF_E_INVALID, "rf_conv3d_k3_gn: no fused statistics / pooling without the ReLU");
        if (edge >= 8) return conv_use_big(n, edge, a.cout16) ? dispatch_nb<8, 8, 8, 1, TILE_BIG>(a, s) : dispatch_nb<4, 4, 8, 1, TILE_SMALL>(a, s);
        return edge == 4 ? dispatch_nb<4, 4, 4, 2, TILE_SMALL>(a, s) : dispatch_nb<2, 2, 2, 16, TILE_SMALL>(a, s);
    }
    if (conv_use_cin1(c0, c1, edge, cout)) return cout == 8 ? launch_cin1<8>(a, s) : launch_cin1<6>(a, s);
    // whole 4^3 / 2^3 volumes: the position-major kernel (conv3d_small.hip) leaves out every zero-padding tap
    if ((!pool_mode || edge == 4) && rf_conv3_small_takes(c0, c1, n, edge, cout))
        return rf_conv3_small_launch(src0, c0, n, edge, gn_affine, w_packed, cout, out, stats, stream, pool_out, pool_stats);
    // 512-voxel workgroup tiles when that still gives the 256 CUs a few workgroups each; otherwise 128-voxel tiles
    const bool big = conv_use_big(n, edge, a.cout16);
    if (edge >= 8) return big ? dispatch_nb<8, 8, 8, 1, TILE_BIG>(a, s) : dispatch_nb<4, 4, 8, 1, TILE_SMALL>(a, s);
    // 4^3 / 2^3 volumes: always the 128-voxel tiles (2 / 16 whole samples per workgroup); the 512-voxel multi-sample forms
    // need 5-16 halo rows per thread in registers, spill, and measured slower
    if (edge == 4) return n >= 4096 ? dispatch_nb<4, 4, 4, 4, TILE_MID>(a, s) : dispatch_nb<4, 4, 4, 2, TILE_SMALL>(a, s);
    return dispatch_nb<2, 2, 2, 16, TILE_SMALL>(a, s);
}

extern "C" int rf_conv3d_k3_gn_relu(const float* src0, int c0, const float* src1, int c1, int n, int edge,
                                    const float* gn_affine, const float* w_packed, int cout,
                                    float* out, void* stream) {
    return conv3d_impl(src0, c0, src1, c1, n, edge, gn_affine, w_packed, cout, out, nullptr, stream);
}

// the same convolution with the ReLU optional: relu = 0 is the data-gradient conv of the backward pass (rfuse/autograd.py)
extern "C" int rf_conv3d_k3_gn(const float* src0, int c0, const float* src1, int c1, int n, int edge,
                               const float* gn_affine, const float* w_packed, int cout, int relu, float* out, void* stream) {
    return conv3d_impl(src0, c0, src1, c1, n, edge, gn_affine, w_packed, cout, out, nullptr, stream, nullptr, nullptr, 0, relu != 0);
}

// fused MaxPool3d(2): only the 8^3-box tiling (edge >= 8, enough boxes, not the cin == 1 kernel) holds whole pooling cells
extern "C" int rf_conv3d_pool_supported(int c0, int c1, int n, int edge, int cout) {
    if (edge == 4) return rf_conv3_small_takes(c0, c1, n, edge, cout);      // position-major kernel pools from its epilogue tile
    return edge >= 8 && rf_is_pow2(edge) && edge <= 128 && !conv_use_cin1(c0, c1, edge, cout) && conv_use_big(n, edge, rf_round_up(cout, 16));
}

extern "C" int rf_conv3d_k3_gn_relu_pool(const float* src0, int c0, const float* src1, int c1, int n, int edge,
                                         const float* gn_affine, const float* w_packed, int cout,
                                         float* out, double* stats, float* pool_out, double* pool_stats, void* stream) {
    RF_REQUIRE(pool_out, RF_E_INVALID, "rf_conv3d_k3_gn_relu_pool: null pooled output");
    RF_REQUIRE(rf_conv3d_pool_supported(c0, c1, n, edge, cout), RF_E_UNSUPPORTED,
               "rf_conv3d_k3_gn_relu_pool: shape not on the 8^3-box tiling (ask rf_conv3d_pool_supported)");
    RF_REQUIRE(out || !stats, RF_E_INVALID, "rf_conv3d_k3_gn_relu_pool: statistics of an output that is not written");
    return conv3d_impl(src0, c0, src1, c1, n, edge, gn_affine, w_packed, cout, out, stats, stream, pool_out, pool_stats, out ? 1 : 2);
}

extern "C" int rf_conv3d_k3_gn_relu_stats(const float* src0, int c0, const float* src1, int c1, int n, int edge,
                                          const float* gn_affine, const float* w_packed, int cout,
                                          float* out, double* stats, void* stream) {
    RF_REQUIRE(stats, RF_E_INVALID, "rf_conv3d_k3_gn_relu_stats: null stats buffer");
    return conv3d_impl(src0, c0, src1, c1, n, edge, gn_affine, w_packed, cout, out, stats, stream);
}
