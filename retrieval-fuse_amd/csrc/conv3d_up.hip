// rf_conv3d_up_k3_gn_relu: the DECODER form of SingleConv 'gcr' (reference model/unet.py:19-76 with the nearest x2
// upsample + concat of :297-308 / :354-360) for gfx950 -- same result as rf_conv3d_k3_gn_relu on (skip, low-res) sources,
// with the upsampled channels convolved in LOW resolution.
//
// A 3x3x3 convolution over a nearest-x2-upsampled volume reads, per axis, only TWO distinct low-res voxels:
//     output x even (phase 0): taps x-1 | x, x+1  ->  low-res X-1 (weight w0)      and X   (w1 + w2),  X = x >> 1
//     output x odd  (phase 1): taps x-1, x | x+1  ->  low-res X   (weight w0 + w1) and X+1 (w2)
// so for each of the 8 output parities (pz,py,px) the 27 taps collapse to 2x2x2 taps with pre-summed weights: 8/27 of
// the multiply-adds and a (T/2+2)^3 instead of a (T+2)^3 halo for those channels.  Zero padding commutes (the upsampled
// border is padded with zeros exactly where the low-res border is), GroupNorm is per (sample, channel) and commutes too.
// The retrieval backbone's decoders spend 2/3 of their channels on the upsampled source (192->64 @4^3: 128 of 192,
// 96->56 @8^3: 64 of 96), the final decoder's first conv all of them.
//
// Work split: all voxels of one MFMA m-block must share their B operand, i.e. their parity.  A workgroup of 8 waves owns
// a T^3 box (T = 8: one box of a sample; T = 4: four whole 4^3 samples); WAVE w OWNS PARITY w = (pz,py,px): its MB
// m-blocks are the (T/2)^3 lattice of that parity.  K loop:
//   A) skip channels (c0): 4-channel chunks, 27 taps on the full-res halo box, weight slab [27][4][NB*16] shared by the
//      8 waves through LDS (LDS-DMA, double buffered) -- the scheme of conv3d_mfma.hip.
//   B) upsampled channels (c1): 8-channel chunks, 8 taps on the low-res halo box (double buffered in LDS); every wave
//      needs its own parity's weights [8 taps][8 ch][NB*16], read once per workgroup: they stream by LDS-DMA through a
//      wave-private ring, several k-steps ahead of the MFMAs.  For 8^3 boxes the halo rows of the next chunk are LDS-DMA'd
//      too (staging area, normalised into the other box buffer mid-chunk): the phase is one software pipeline without a
//      s_waitcnt vmcnt(0) or a VGPR-staged load in it (see the comment at the phase).
// All per-lane index arithmetic of the DMAs / row loads is done once per phase, per chunk only scalar bases change: VALU
// instructions issue through the same port as the MFMAs, and the round-1 form (indices recomputed per piece to save
// registers) cost 6 % of the kernel.
// Epilogue: accumulators -> LDS tile [16 cout][box] -> ReLU -> contiguous float4 rows to HBM (the parity split would
// otherwise leave every lane with stride-2 scalars), GroupNorm statistics of the output for the next layer as in
// conv3d_mfma.hip (float64, fixed order, per workgroup tile).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((address_space(1))) const void* rf_gptr;
typedef __attribute__((address_space(3))) void* rf_lptr;

// ---------------------------------------------------------------------------------------------------- weight image
// [27][c0_4][cout16] (skip channels, as rf_conv3_pack_weight)  ++  [c1_8/8][8 parities][8 taps][8 ch][cout16]
static inline size_t up_c0_floats(int cout, int c0) { return (size_t)27 * rf_round_up(c0, 4) * rf_round_up(cout, 16); }

extern "C" size_t rf_conv3_up_packed_floats(int cout, int c0, int c1) {
    return up_c0_floats(cout, c0) + (size_t)rf_round_up(c1, 8) * 64 * rf_round_up(cout, 16);
}

__global__ void k_conv3_up_pack(const float* __restrict__ w, int cout, int c0, int c1, int c0_4, int c1_8, int cout16, float* __restrict__ wp) {
    const int cin = c0 + c1;
    const size_t n0 = (size_t)27 * c0_4 * cout16, total = n0 + (size_t)c1_8 * 64 * cout16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < n0) {
            const int co = (int)(i % cout16), ci = (int)((i / cout16) % c0_4), tap = (int)(i / ((size_t)cout16 * c0_4));
            if (co < cout && ci < c0) v = w[((size_t)co * cin + ci) * 27 + tap];
        } else {
            const size_t q = i - n0;
            const int co = (int)(q % cout16), k8 = (int)((q / cout16) % 8), tap = (int)((q / ((size_t)cout16 * 8)) % 8);
            const int ph = (int)((q / ((size_t)cout16 * 64)) % 8), chunk = (int)(q / ((size_t)cout16 * 512));
            const int ci = chunk * 8 + k8;
            if (co < cout && ci < c1) {
                // per axis: parity 0: low-res tap 0 <- {0}, tap 1 <- {1,2};  parity 1: tap 0 <- {0,1}, tap 1 <- {2}
                const int pz = ph >> 2, py = (ph >> 1) & 1, px = ph & 1, tz = tap >> 2, ty = (tap >> 1) & 1, tx = tap & 1;
                const int z_lo = tz == 0 ? 0 : (pz ? 2 : 1), z_hi = tz == 0 ? (pz ? 1 : 0) : 2;
                const int y_lo = ty == 0 ? 0 : (py ? 2 : 1), y_hi = ty == 0 ? (py ? 1 : 0) : 2;
                const int x_lo = tx == 0 ? 0 : (px ? 2 : 1), x_hi = tx == 0 ? (px ? 1 : 0) : 2;
                const float* wk = w + ((size_t)co * cin + c0 + ci) * 27;
                double s = 0.0;                                  // summed in float64, rounded once
                for (int dz = z_lo; dz <= z_hi; ++dz)
                    for (int dy = y_lo; dy <= y_hi; ++dy)
                        for (int dx = x_lo; dx <= x_hi; ++dx) s += (double)wk[(dz * 3 + dy) * 3 + dx];
                v = (float)s;
            }
        }
        wp[i] = v;
    }
}

extern "C" int rf_conv3_up_pack_weight(const float* w_oidhw, int cout, int c0, int c1, float* w_packed, void* stream) {
    RF_REQUIRE(w_oidhw && w_packed && cout > 0 && c0 >= 0 && c1 > 0, RF_E_INVALID, "rf_conv3_up_pack_weight: bad arguments");
    const size_t total = rf_conv3_up_packed_floats(cout, c0, c1);
    const size_t want = (total + 255) / 256;
    hipLaunchKernelGGL(k_conv3_up_pack, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, (hipStream_t)stream, w_oidhw, cout, c0, c1,
                       rf_round_up(c0, 4), rf_round_up(c1, 8), rf_round_up(cout, 16), w_packed);
    RF_CHECK_LAUNCH("rf_conv3_up_pack_weight");
    return RF_OK;
}

// ---------------------------------------------------------------------------------------------------------- kernel
struct UpArgs {
    const float* src0;
    const float* src1;
    const float4* affine;   // GroupNorm per (sample, input channel): (center, scale, shift, -) -> y = (x - center) * scale + shift
    const float* wp;
    float* out;
    int c0, c1, n, edge, cout, c0_4, c1_8, cout16;
    double2* stats;    // optional [n][cout][stats_tiles]
    int stats_tiles;
};

// s_waitcnt vmcnt(n) for an n that is a constant after unrolling (the instruction takes an immediate)
__device__ __forceinline__ void rf_wait_vm(int n) {
#define RF_WVM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        RF_WVM(0) RF_WVM(1) RF_WVM(2) RF_WVM(3) RF_WVM(4) RF_WVM(5) RF_WVM(6) RF_WVM(7) RF_WVM(8) RF_WVM(9) RF_WVM(10) RF_WVM(11) RF_WVM(12)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef RF_WVM
}

// In place: r[m] row c  ->  r[c] row m, rows = the four 16-lane groups of a wave (probed on gfx950, tools/micro/mfma4x4_probe.hip:
// v_permlane32_swap(a, b) exchanges a.rows{2,3} with b.rows{0,1}; v_permlane16_swap(a, b) exchanges a.row1 <-> b.row0, a.row3 <-> b.row2).
// Turns the four A operands of v_mfma_f32_16x16x4_f32 (lane = (channel, voxel % 16) per m-block) into the four A operands of
// v_mfma_f32_4x4x1_16b_f32 (lane = voxel, one register per channel).
__device__ __forceinline__ void rf_rows4_transpose(float& r0, float& r1, float& r2, float& r3) {
    auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(r0), __float_as_uint(r2), false, false);
    auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(r1), __float_as_uint(r3), false, false);
    auto p01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
    auto p23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
    r0 = __uint_as_float(p01[0]); r1 = __uint_as_float(p01[1]); r2 = __uint_as_float(p23[0]); r3 = __uint_as_float(p23[1]);
}

// lane id from nothing (v_mbcnt): a value the register allocator can recompute instead of keeping threadIdx.x alive -- or spilling it --
// across the MFMA phases
__device__ __forceinline__ int rf_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ unsigned up_xcd_contiguous(unsigned b, unsigned g) {      // see conv3d_mfma.hip
    const unsigned per = g >> 3, rem = g & 7u, k = b & 7u;
    return k * per + (k < rem ? k : rem) + (b >> 3);
}

// Order of a wave's parity lattice, v = mb*16 + i -> (sample, Z, Y, X).  The lattice plane Z is constant inside an m-block
// (Z = mb): an m-block on the first / last plane then reads only zero padding through its z-border taps.
//   T = 8 (one box, 4^3 lattice):      v = Z*16 + Y*4 + X
//   T = 4 (four samples, 2^3 lattice): v = Z*16 + s*4 + Y*2 + X
template <int TE>
__device__ __forceinline__ void up_lattice(int v, int& s, int& Z, int& Y, int& X) {
    if (TE == 8) { X = v & 3; Y = (v >> 2) & 3; Z = v >> 4; s = 0; }
    else { X = v & 1; Y = (v >> 1) & 1; s = (v >> 2) & 3; Z = v >> 4; }
}

template <int TE, int SPW, int MB, int NB>
struct UpTile {
    static constexpr int NW = 8, NT = 512;
    static constexpr int L = TE / 2, LH = L + 2, HE = TE + 2;
    // x-row stride of the full-res halo box.  T = 8: the 16 lattice points of an A read are 2 floats apart in x and 2 rows apart in y;
    // with rows of 10 floats they fall on 14 of the 32 ds_read_b32 banks (2-way) and the second channel of the 32-lane group
    // (stride 1000) lands on the same even banks (3x the conflict-free cycles, SQ_LDS_BANK_CONFLICT / IDX_ACTIVE 0.43).  Rows of 12
    // floats spread them over all 16 even banks, an ODD channel stride puts the other channel on the odd banks: conflict-free.
    static constexpr int HXA = TE == 8 ? 12 : HE;
    static constexpr int CH0 = HE * HE * HXA + (TE == 8 ? 1 : 0), CH1 = LH * LH * LH;
    static constexpr int XS0 = SPW * 4 * CH0;                          // full-res halo box of a 4-channel chunk
    static constexpr int XS1 = SPW * 8 * CH1;                          // low-res halo box of an 8-channel chunk (x2 buffers)
    static constexpr int XS = ((XS0 > 2 * XS1 ? XS0 : 2 * XS1) + 3) / 4 * 4;
    static constexpr int NCO = NB * 16;
    static constexpr int WSLAB = 27 * 4 * NCO;
    static constexpr int WSLAB_PAD = (WSLAB + 255) / 256 * 256;
    static constexpr int P = SPW * TE * TE * TE;
    static constexpr int EPI = 16 * (P + 1);                           // epilogue tile [16 cout][P + 1]
    static constexpr int MAIN = XS + 2 * WSLAB_PAD;                     // phase A: halo box + two shared weight slabs
    static constexpr int RING = 7;                                     // TE == 8: 1-KiB pieces of weights per wave-private ring
    static constexpr int MAINB = TE == 8 ? 2 * XS1 + NW * RING * 256 + 288 * 6 + 32      // two low-res boxes + rings + row staging
                                         : 2 * XS1 + NW * 2 * 16 * NCO;                  // two low-res boxes + per-wave quarter slabs x2
    static constexpr int LDS_FLOATS = MAIN > MAINB ? (MAIN > EPI ? MAIN : EPI) : (MAINB > EPI ? MAINB : EPI);
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);
    static_assert(LDS_BYTES <= 81920, "two workgroups per CU");
    static_assert(P == NW * MB * 16, "8 waves x MB x 16 voxels must cover the box");
    static_assert(SPW * L * L * L == MB * 16, "one parity lattice per wave");
    static_assert((TE == 8 && SPW == 1) || (TE == 4 && SPW == 4), "built tiles: one 8^3 box, or four whole 4^3 samples");
};

// HALF (NB = 4, 48 < cout <= 56 -- the retrieval backbone's 96 -> 56 conv): couts 48..55 are NOT a fourth, half-empty 16-wide n-block
// (1/8 of all MFMA cycles multiplying zero columns) but two 4-cout groups on v_mfma_f32_4x4x1_16b_f32: 16 blocks = the wave's 16
// groups of 4 voxels, A = one register per channel with lane = voxel (rf_rows4_transpose of the four m-blocks' A operands, after
// their last 16x16x4 use), B = w[channel][48 + 4h + lane % 4], D[reg i][lane 4b + j] = out(voxel 4b + i, cout 48 + 4h + j).
// Same flop rate, no padding: 8 x 8 cycles instead of 4 x 32 per k-step, and 8 accumulator VGPRs instead of 16.  The z-border
// skip does not apply to these (an instruction covers all four m-blocks; the skipped taps add exact zeros either way).
template <int TE, int SPW, int MB, int NB, bool HALF = false>
__global__ __launch_bounds__(512, 4) void k_conv3_up(UpArgs a) {
    static_assert(!HALF || (NB == 4 && MB == 4 && TE == 8), "the half n-block form is built for the 8^3 tile with 64-wide slabs");
    constexpr int NBF = HALF ? NB - 1 : NB;                              // full 16-cout n-blocks
    using T = UpTile<TE, SPW, MB, NB>;
    constexpr int L = T::L, LH = T::LH, HE = T::HE, HXA = T::HXA, CH0 = T::CH0, CH1 = T::CH1, NCO = T::NCO, NT = T::NT, P = T::P;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* wsb = smem + T::XS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pz = wave >> 2, py = (wave >> 1) & 1, px = wave & 1;     // this wave's output parity
    const int edge = a.edge, half = edge >> 1, cin = a.c0 + a.c1;

    // ---- which box?
    int n0, z0 = 0, y0 = 0, x0 = 0, tile = 0;
    if (SPW == 1) {
        const unsigned lb = gridDim.y == 1 ? up_xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
        const int tpe = edge / TE;
        int t = (int)lb;
        tile = t % (tpe * tpe * tpe);
        x0 = (t % tpe) * TE; t /= tpe;
        y0 = (t % tpe) * TE; t /= tpe;
        z0 = (t % tpe) * TE; t /= tpe;
        n0 = t;
    } else {
        n0 = blockIdx.x * SPW;
    }
    const int cob = blockIdx.y * NCO;

    constexpr int ROT = NCO >= 32 ? 16 : 0;                             // slab bank rotation, as conv3d_mfma.hip

    // Everything from here on is instantiated per z-border variant of this wave: an m-block whose voxels lie on the first
    // (LO) / last (HI) z slice of the VOLUME reads nothing but zero padding through its dz = -1 / +1 taps (phase A) and its
    // low-res tz = 0 / 1 taps (phase B) -- those MFMAs add exact zeros and are left out (same result bit for bit).  With
    // The lattice plane Z of m-block mb is mb (up_lattice), so only mb 0 of the pz = 0 waves / the last mb of the pz = 1 waves qualify.
    // Whole-kernel variants (not a test inside the loop) keep the accumulators in place; barriers match across variants.
    auto run = [&](auto lo_c, auto hi_c) {
    constexpr unsigned LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
    f32x4 acc[MB][NBF];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBF; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 acc4[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};      // HALF: couts 48 + 4h + (lane % 4) of voxels 4 (lane / 4) + i
    // HALF: the 8 MFMAs of a k-step on couts 48..55; b4[c] = (h = 0, h = 1) weights of channel c.  Destroys a0..a3.
    auto mfma_half = [&](float& a0, float& a1, float& a2, float& a3, const float2 (&b4)[4]) {
        rf_rows4_transpose(a0, a1, a2, a3);
        const float at[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc4[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(at[c], b4[c].x, acc4[0], 0, 0, 0);
            acc4[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(at[c], b4[c].y, acc4[1], 0, 0, 0);
        }
    };

    // ================================================================================= A) skip channels, 27 taps
    if (a.c0 > 0) {
        // per-lane LDS offsets of the operands: voxel v = mb*16 + li of this wave's parity lattice, channel kq.  Derived
        // from an opaque copy of the lane id so that they only live inside this phase of the kernel.
        int lane_a = lane;
        asm volatile("" : "+v"(lane_a));
        const int kq_a = lane_a >> 4, li_a = lane_a & 15;
        int aoff0[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int v = mb * 16 + li_a;
            int X, Y, Z, s;
            up_lattice<TE>(v, s, Z, Y, X);
            aoff0[mb] = (s * 4 + kq_a) * CH0 + ((2 * Z + pz) * HE + (2 * Y + py)) * HXA + (2 * X + px);
        }
        int boff[NBF];
#pragma unroll
        for (int nb = 0; nb < NBF; ++nb) boff[nb] = kq_a * NCO + ((nb * 16 + li_a + ROT * (kq_a & 1)) % NCO);
        const int l4_a = lane_a & 3;                                    // HALF: slab row k = tap*4 + c holds cout 48 + 4h + j at (48 + ROT (c & 1)) % NCO + 4h + j
        // Weight slab DMA: 1-KiB pieces of RPP slab rows [tap][ci] x NCO; piece q is rows [q RPP, (q+1) RPP).  Everything that
        // depends on the lane is computed ONCE (the per-chunk part is a scalar base): the per-piece index arithmetic of the
        // round-1 form -- a runtime modulo and 64-bit multiplies among it -- ran on the VALU port the MFMAs issue through.
        constexpr int LPR = NCO / 4, RPP = 64 / LPR;                    // lanes per slab row; slab rows per piece (4, 8, 16)
        constexpr int NPIECE = (27 * 4 + RPP - 1) / RPP;
        const int wr_a = lane_a / LPR, wslot_a = (lane_a % LPR) * 4;
        const int wcol_a = (wslot_a + NCO - ROT * (wr_a & 1)) % NCO;    // slab row parity = piece row parity (RPP is even)
        int wco_a = cob + wcol_a;
        if (wco_a >= a.cout16) wco_a = wcol_a % a.cout16;               // block wider than the image: masked at the store
        const int wtap_a = wr_a >> 2;                                   // tap of this lane's row inside a piece
        const unsigned walane = 4u * (unsigned)((wtap_a * a.c0_4 + (wr_a & 3)) * a.cout16 + wco_a);   // bytes
        auto dma_weights = [&](int cbase, int buf) {
            float* dst = wsb + buf * T::WSLAB_PAD;
            unsigned off = walane;                                      // opaque: SGPR base + 32-bit lane offset addressing
            asm volatile("" : "+v"(off));
#pragma unroll
            for (int i = 0; i < (NPIECE + 7) / 8; ++i) {
                const int q = wave + i * 8;
                if (q < NPIECE) {
                    const int tap0 = q * (RPP / 4);
                    const char* src = reinterpret_cast<const char*>(a.wp + ((size_t)tap0 * a.c0_4 + cbase) * a.cout16);
                    if (RPP == 4 || tap0 + wtap_a < 27)                 // rows past tap 26 of the last piece are never read
                        __builtin_amdgcn_global_load_lds((rf_gptr)(src + off), (rf_lptr)(dst + q * 256), 16, 0, 0);
                }
            }
        };
        // Halo rows of the chunk: thread r -> row (sample s, chunk channel c, hz, hy) of TE + 2 floats, through registers (they
        // are normalised on the way into LDS).  Per-lane constants once, per chunk a scalar base.
        constexpr int ROWS = SPW * 4 * HE * HE;
        constexpr int RPT = (ROWS + NT - 1) / NT;
        float xraw[RPT][TE + 2];
        float xce[RPT], xsc[RPT], xsh[RPT];
        const bool has_l = x0 > 0, has_r = x0 + TE < edge;
        unsigned rowoffA[RPT], affoffA[RPT];                            // bytes behind (sample n0, channel cbase) of src0 / the affine table
        int rowboxA[RPT], rowcA[RPT];                                   // LDS row; chunk channel (>= 1 << 20: zero row, < 0: no row)
        {
            int tid_a = tid;
            asm volatile("" : "+v"(tid_a));
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int r = tid_a + i * NT;
                const int hy = r % HE, hz = (r / HE) % HE, c = (r / (HE * HE)) % 4, s = r / (HE * HE * 4);
                const int z = z0 + hz - 1, y = y0 + hy - 1;
                const bool in = r < ROWS && n0 + s < a.n && (unsigned)z < (unsigned)edge && (unsigned)y < (unsigned)edge;
                rowoffA[i] = in ? 4u * (unsigned)((((s * a.c0 + c) * edge + z) * edge + y) * edge + x0) : 4u * (unsigned)x0;
                affoffA[i] = in ? 16u * (unsigned)(s * cin + c) : 0u;
                rowboxA[i] = (s * 4 + c) * CH0 + (hz * HE + hy) * HXA;
                rowcA[i] = r < ROWS ? (in ? c : (1 << 20)) : -1;
            }
        }
        auto issue_rows = [&](int cbase) {
            const char* vol0 = reinterpret_cast<const char*>(a.src0 + ((size_t)n0 * a.c0 + cbase) * edge * edge * edge);
            const char* aff0 = reinterpret_cast<const char*>(a.affine + ((size_t)n0 * cin + cbase));
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                if (rowcA[i] >= 0 && cbase + rowcA[i] < a.c0) {
                    unsigned ro = rowoffA[i], ao = affoffA[i];
                    asm volatile("" : "+v"(ro), "+v"(ao));
                    { const float4 af = *reinterpret_cast<const float4*>(aff0 + ao); xce[i] = af.x; xsc[i] = af.y; xsh[i] = af.z; }
                    const float* row = reinterpret_cast<const float*>(vol0 + ro);
#pragma unroll
                    for (int q = 0; q < TE / 4; ++q) {
                        const float4 t = reinterpret_cast<const float4*>(row)[q];
                        xraw[i][1 + 4 * q] = t.x; xraw[i][2 + 4 * q] = t.y; xraw[i][3 + 4 * q] = t.z; xraw[i][4 + 4 * q] = t.w;
                    }
                    if (has_l) xraw[i][0] = row[-1];
                    if (has_r) xraw[i][TE + 1] = row[TE];
                }
            }
        };
        auto commit_rows = [&](int cbase) {
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                if (rowcA[i] >= 0) {
                    float v[TE + 2];
                    if (cbase + rowcA[i] < a.c0) {
                        const float ce = xce[i], sc = xsc[i], sh = xsh[i];
#pragma unroll
                        for (int j = 1; j <= TE; ++j) v[j] = fmaf(xraw[i][j] - ce, sc, sh);
                        v[0] = has_l ? fmaf(xraw[i][0] - ce, sc, sh) : 0.f;
                        v[TE + 1] = has_r ? fmaf(xraw[i][TE + 1] - ce, sc, sh) : 0.f;
                    } else {
#pragma unroll
                        for (int j = 0; j < TE + 2; ++j) v[j] = 0.f;
                    }
                    float* dst = xs + rowboxA[i];
#pragma unroll
                    for (int j = 0; j < TE + 2; ++j) dst[j] = v[j];
                }
            }
        };

        dma_weights(0, 0);
        issue_rows(0);
        commit_rows(0);
        __syncthreads();
        int buf = 0;
        for (int cbase = 0; cbase < a.c0; cbase += 4) {
            const bool more = cbase + 4 < a.c0;
            if (more) {
                issue_rows(cbase + 4);
                dma_weights(cbase + 4, buf ^ 1);
            }
            {
                const float* ws = wsb + buf * T::WSLAB_PAD;
                float av[2][MB], bv[2][NBF];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) av[0][mb] = xs[aoff0[mb]];
#pragma unroll
                for (int nb = 0; nb < NBF; ++nb) bv[0][nb] = ws[boff[nb]];
#pragma unroll
                for (int t = 0; t < 27; ++t) {
                    const int cur = t & 1, nxt = cur ^ 1;
                    // Software pipeline, pinned with scheduling barriers (left alone, hipcc sinks the reads of tap t+1 below the MFMAs
                    // of tap t and every tap starts with a full LDS round trip): m-block 1 of tap t | reads of tap t+1 interleaved
                    // with the other m-blocks of tap t.  The compiler's s_waitcnt lgkmcnt(0) then lands in front of the first MFMA of tap t+1,
                    // 3/4 of a tap (~400 cycles) after the reads went out.
                    auto mfma_mb = [&](int mb) {
                        // m-block on the first / last z slice of the volume: its dz = -1 / +1 taps read only zero padding
                        if ((t / 9 == 0 && ((LO >> mb) & 1u)) || (t / 9 == 2 && ((HI >> mb) & 1u))) return;        // compile-time
#pragma unroll
                        for (int nb = 0; nb < NBF; ++nb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][mb], bv[cur][nb], acc[mb][nb], 0, 0, 0);
                    };
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_mb(1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (t + 1 < 27) {
                        const int t1 = t + 1;
                        const int toff = ((t1 / 9) * HE + (t1 / 3) % 3) * HXA + t1 % 3;
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) av[nxt][mb] = xs[aoff0[mb] + toff];
#pragma unroll
                        for (int nb = 0; nb < NBF; ++nb) bv[nxt][nb] = ws[boff[nb] + t1 * 4 * NCO];
                    }
                    float2 b4[4];                                       // HALF: this tap's weights of couts 48..55, used at the end of the step
                    if constexpr (HALF) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float* w4 = ws + (t * 4 + c) * NCO + (48 + ROT * (c & 1)) % NCO + l4_a;
                            b4[c] = make_float2(w4[0], w4[4]);
                        }
                    }
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
                        if (mb != 1) mfma_mb(mb);
                    // one LDS read behind every MFMA: the wave's instruction stream stays MFMA-dense while the reads go out (-1 %)
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    if constexpr (HALF) {
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_half(av[cur][0], av[cur][1], av[cur][2], av[cur][3], b4);
                    }
                }
            }
            __syncthreads();
            if (more) commit_rows(cbase + 4);
            __syncthreads();
            buf ^= 1;
        }
    }

    // ========================================================================== B) upsampled channels, 8 low-res taps
    if constexpr (TE == 8) {
        // One 8^3 box (L = 4): low-res halo rows are 4 floats = one 16-byte LDS-DMA unit.  Nothing of the next chunk passes
        // through VGPRs and no address is recomputed per step (in the round-1 form the halo rows lived in 9 VGPRs across the
        // MFMA steps -- hipcc spilled them right behind their loads, s_waitcnt vmcnt(0) included -- and every weight DMA redid
        // ~45 VALU instructions of index arithmetic, quarter-rate multiplies among them, competing with the MFMAs for the
        // SIMD's issue port):
        //   * wave w stages channel w of the chunk: lanes 0..35 DMA the 6 x 6 (z, y) rows, their two x-halo floats and the
        //     channel's GroupNorm triple into a staging area; at mid-chunk the same lanes normalise them into the other box
        //     buffer (zero padding outside the volume).  The wave that DMAs a row commits it: no barrier in between.
        //   * this wave's weights [tap][k8][cout16] stream through a wave-private ring of RING 1-KiB pieces (one full-wave DMA
        //     instruction = GS k-steps of 4 rows x NCO floats), RING - 2 pieces ahead of the MFMAs.  The number of DMA
        //     instructions between any two points of the loop is static (pieces past the end are re-reads of the last one)
        //     and vmcnt retires in order, so "piece P+1 has landed" is s_waitcnt vmcnt(RING - 3 [+ RDMA while the chunk's row
        //     DMAs are younger than that piece]) and "the rows have landed" is vmcnt(pieces issued since).
        //   * one software pipeline across the whole phase: operand reads of k-step S+1 go out in the middle of k-step S, also
        //     across the chunk boundary (one LDS-only barrier per chunk, no vmcnt(0) anywhere in the loop).
        static_assert(L == 4 && LH == 6 && SPW == 1, "TE == 8 tile");
        constexpr int RING = T::RING, STEP = 4 * NCO;                  // ring pieces per wave; floats per k-step
        constexpr int GS = 256 / STEP, PPC = 16 / GS;                   // k-steps per piece (1, 2, 4); pieces per chunk
        constexpr int CS = 8;                                           // the k-step in whose middle the next chunk's rows are committed
        constexpr int RDMA = 4;                                         // row, left halo, right halo, affine: DMA instructions per chunk
        float* const stg = smem + 2 * T::XS1 + T::NW * RING * 256;        // [288][4] raw rows | [288] left | [288] right | [8][4] affine
        float* const stgl = stg + 288 * 4;
        float* const stgr = stgl + 288;
        float* const stga = stgr + 288;
        float* const wslab = smem + 2 * T::XS1 + wave * (RING * 256);
        const int Z0 = z0 >> 1, Y0 = y0 >> 1, X0 = x0 >> 1;
        const bool has_l = X0 > 0, has_r = X0 + L < half;
        const int nstep = (a.c1_8 >> 3) * 16, npiece = (a.c1_8 >> 3) * PPC;
        const size_t hvol = (size_t)half * half * half;

        const int lane_b = rf_lane();
        const int kq = lane_b >> 4, li = lane_b & 15;
        // ---- staging: row (hz, hy) = lane (< 36) of channel `wave`.  Recomputed from an opaque copy of the lane id where it is
        //      used (once per chunk, ~10 VALU): held across the loop these values get spilled, and a reload is a s_waitcnt vmcnt(0).
        auto row_of_lane = [&](int l, bool& in, unsigned& off) {
            const int hz = l / 6, hy = l - hz * 6;
            const int rz = Z0 + hz - 1, ry = Y0 + hy - 1;
            in = l < 36 && (unsigned)rz < (unsigned)half && (unsigned)ry < (unsigned)half;
            // BYTES from the channel's volume (SGPR base + 32-bit lane offset addressing); rows outside the volume read row (0, 0) at
            // X0 (their -4 / +16 halo reads stay inside the row: the offset is unsigned, a wrap would be +4 GiB)
            off = 4u * (unsigned)((in ? (rz * half + ry) * half : 0) + X0);
        };
        // ---- per-lane constant of the weight DMA: row r = lane / (NCO/4) of the k-step, bank-rotated column
        const int wr = lane_b / (NCO / 4), wslot = (lane_b % (NCO / 4)) * 4;
        const int wcol = (wslot + NCO - (16 * (wr & 3)) % NCO) % NCO;
        int wco = cob + wcol;
        if (wco >= a.cout16) wco = wcol % a.cout16;                     // block wider than the image: masked at the store
        const unsigned wlane = 4u * (unsigned)(wr * a.cout16 + wco);    // bytes
        const float* const wimg = a.wp + (size_t)27 * a.c0_4 * a.cout16 + (size_t)wave * 64 * a.cout16;

        int aoff1[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int v = mb * 16 + li;
            int X, Y, Z, s;
            up_lattice<TE>(v, s, Z, Y, X);
            aoff1[mb] = kq * CH1 + ((Z + pz) * LH + (Y + py)) * LH + (X + px);
        }
        int boff1[NBF];
#pragma unroll
        for (int nb = 0; nb < NBF; ++nb) boff1[nb] = kq * NCO + ((nb * 16 + li + 16 * kq) % NCO);
        const int l4_b = lane_b & 3;                                    // HALF: ring row c holds cout 48 + 4h + j at (48 + 16 c) % NCO + 4h + j

        auto dma_piece = [&](int P, int slot) {                         // image rows [4 GS P, 4 GS (P + 1)) of this parity -> ring slot
            if (P >= npiece) P = npiece - 1;                            // past the end: keeps the DMA count static, lands in a dead slot
            const float* src = wimg + ((size_t)(P / PPC) * 512 + (size_t)(P % PPC) * (4 * GS)) * a.cout16;
            unsigned off = wlane;                                       // opaque: SGPR base + 32-bit lane offset at the instruction, not a
            asm volatile("" : "+v"(off));                               // hoisted 64-bit address pair per lane
            __builtin_amdgcn_global_load_lds((rf_gptr)(reinterpret_cast<const char*>(src) + off), (rf_lptr)(wslab + slot * 256), 16, 0, 0);
        };
        auto dma_rows = [&](int cbase) {                                // channel cbase + wave -> staging (RDMA instructions)
            int ci = cbase + wave;
            if (ci >= a.c1) ci = a.c1 - 1;                              // padded channel: staged, committed as zeros
            const char* vol1 = reinterpret_cast<const char*>(a.src1 + ((size_t)n0 * a.c1 + ci) * hvol);
            int l = rf_lane();                                          // opaque: not hoisted out of the loop (and then spilled)
            asm volatile("" : "+v"(l));
            bool in;
            unsigned off;
            row_of_lane(l, in, off);
            if (l < 36) {
                __builtin_amdgcn_global_load_lds((rf_gptr)(vol1 + off), (rf_lptr)(stg + wave * 36 * 4), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((rf_gptr)(vol1 + (off - (has_l ? 4u : 0u))), (rf_lptr)(stgl + wave * 36), 4, 0, 0);
                __builtin_amdgcn_global_load_lds((rf_gptr)(vol1 + (off + (has_r ? 4u * L : 0u))), (rf_lptr)(stgr + wave * 36), 4, 0, 0);
            }
            if (l == 0)
                __builtin_amdgcn_global_load_lds((rf_gptr)(a.affine + ((size_t)n0 * cin + a.c0 + ci)), (rf_lptr)(stga + wave * 4), 16, 0, 0);
        };
        auto commit_rows = [&](int cbase, float* box) {                 // staging -> normalised rows of channel `wave`
            int l = rf_lane();                                          // opaque: not hoisted out of the loop (and then spilled)
            asm volatile("" : "+v"(l));
            if (l < 36) {
                bool in;
                unsigned off;
                row_of_lane(l, in, off);
                const float4 af = *reinterpret_cast<const float4*>(stga + wave * 4);
                const float4 u = *reinterpret_cast<const float4*>(stg + (wave * 36 + l) * 4);
                const float ul = stgl[wave * 36 + l], ur = stgr[wave * 36 + l];
                const bool ok = in && cbase + wave < a.c1;
                float2 v0, v1, v2;
                v0.x = ok && has_l ? fmaf(ul - af.x, af.y, af.z) : 0.f;
                v0.y = ok ? fmaf(u.x - af.x, af.y, af.z) : 0.f;
                v1.x = ok ? fmaf(u.y - af.x, af.y, af.z) : 0.f;
                v1.y = ok ? fmaf(u.z - af.x, af.y, af.z) : 0.f;
                v2.x = ok ? fmaf(u.w - af.x, af.y, af.z) : 0.f;
                v2.y = ok && has_r ? fmaf(ur - af.x, af.y, af.z) : 0.f;
                float2* dst = reinterpret_cast<float2*>(box + wave * CH1 + l * LH);
                dst[0] = v0; dst[1] = v1; dst[2] = v2;
            }
        };
        auto lds_barrier = [&]() {                                      // LDS-only: the weight DMAs in flight stay in flight
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        };

        // ---- prologue: RING - 1 pieces in flight, chunk 0 staged and committed
#pragma unroll
        for (int i = 0; i < RING - 1; ++i) dma_piece(i, i);
        dma_rows(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        commit_rows(0, xs);
        lds_barrier();

        float av[2][MB], bv[2][NBF];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) av[0][mb] = xs[aoff1[mb]];
#pragma unroll
        for (int nb = 0; nb < NBF; ++nb) bv[0][nb] = wslab[boff1[nb]];
        int slot = 0;                                                   // ring slot of the current piece (wave-uniform)
        int buf = 0;
        for (int S0 = 0; S0 < nstep; S0 += 16) {
            const bool more = S0 + 16 < nstep;
            const int P0 = (S0 >> 4) * PPC;                             // first piece of this chunk
            const float* xb = xs + buf * T::XS1;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int cur = s & 1, nxt = cur ^ 1;
                const int tz = s >> 3;                                  // low-res z tap of this k-step (tap = s >> 1)
                auto mfma_mb = [&](int mb) {
                    // low-res row Z-1 of the first lattice plane / Z+1 of the last one is zero padding of the volume
                    if ((tz == 0 && ((LO >> mb) & 1u)) || (tz == 1 && ((HI >> mb) & 1u))) return;                  // compile-time
#pragma unroll
                    for (int nb = 0; nb < NBF; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][mb], bv[cur][nb], acc[mb][nb], 0, 0, 0);
                };
                __builtin_amdgcn_sched_barrier(0);
                mfma_mb(1);
                __builtin_amdgcn_sched_barrier(0);
                float2 b4[4];                                           // HALF: this k-step's weights of couts 48..55 (its piece has landed)
                if constexpr (HALF) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float* w4 = wslab + slot * 256 + (s % GS) * STEP + c * NCO + (48 + 16 * c) % NCO + l4_b;
                        b4[c] = make_float2(w4[0], w4[4]);
                    }
                }
                // ---- middle of k-step S0 + s
                const bool last_of_piece = s % GS == GS - 1;            // the next k-step reads the next piece
                int nslot = slot;
                if (last_of_piece) {
                    // piece P+1 landed?  In flight: pieces P+1 .. P+RING-2, and -- if already issued (s >= 1) and younger than piece
                    // P+1 -- the RDMA row DMAs of this chunk.  They went out in the middle of k-step 0, behind piece
                    // P0 + RING - 1 (GS == 1; P0 + RING - 2 otherwise).
                    const int j = s / GS;                               // P = P0 + j
                    const bool rows_younger = s >= 1 && j + 1 <= (GS == 1 ? RING - 1 : RING - 2);
                    if (more && rows_younger) rf_wait_vm(RING - 3 + RDMA); else rf_wait_vm(RING - 3);
                    nslot = slot + 1;
                    if (nslot == RING) nslot = 0;
                }
                if (s == 15 && more) lds_barrier();                     // the next chunk's box is complete (committed at s = CS)
                if (s < 15 || more) {
                    const float* xn = s < 15 ? xb : xs + (buf ^ 1) * T::XS1;
                    const int s1 = (s + 1) & 15, t1 = s1 >> 1, k1 = s1 & 1;
                    const int toff = ((t1 >> 2) * LH + ((t1 >> 1) & 1)) * LH + (t1 & 1) + k1 * 4 * CH1;
                    const float* wn = wslab + nslot * 256 + (s1 % GS) * STEP;
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) av[nxt][mb] = xn[aoff1[mb] + toff];
#pragma unroll
                    for (int nb = 0; nb < NBF; ++nb) bv[nxt][nb] = wn[boff1[nb]];
                }
                if (last_of_piece) {
                    // refill the slot of the PREVIOUS piece: every read of it has been consumed by an MFMA.  (Reads of the current
                    // piece -- HALF's b4 of this k-step -- may still be in the LDS queue.)
                    int fslot = slot - 1;
                    if (fslot < 0) fslot = RING - 1;
                    dma_piece(P0 + s / GS + RING - 1, fslot);
                }
                if (s == 0 && more) dma_rows((S0 >> 1) + 8);
                if (s == CS && more) {
                    int younger = 0;                                    // pieces issued behind the row DMAs so far
#pragma unroll
                    for (int q = 1; q <= CS; ++q) younger += q % GS == GS - 1;
                    rf_wait_vm(younger);
                    commit_rows((S0 >> 1) + 8, xs + (buf ^ 1) * T::XS1);
                }
                slot = nslot;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    if (mb != 1) mfma_mb(mb);
                if constexpr (HALF) {
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_half(av[cur][0], av[cur][1], av[cur][2], av[cur][3], b4);
                }
            }
            buf ^= 1;
        }
        __syncthreads();                                                // every wave is done with the boxes and its ring
    } else
    {
        constexpr int ROWS = SPW * 8 * LH * LH;                         // rows of LH floats
        constexpr int RPT = (ROWS + NT - 1) / NT;
        float xraw[RPT][LH];
        float xce[RPT], xsc[RPT], xsh[RPT];
        const int Z0 = z0 >> 1, Y0 = y0 >> 1, X0 = x0 >> 1;
        const bool has_l = X0 > 0, has_r = X0 + L < half;
        auto row_coords = [&](int r, int cbase, int& s, int& c, int& hz, int& hy, int& nn, int& ci, int& z, int& y) -> bool {
            hy = r % LH; hz = (r / LH) % LH; c = (r / (LH * LH)) % 8; s = r / (LH * LH * 8);
            nn = n0 + s; ci = cbase + c; z = Z0 + hz - 1; y = Y0 + hy - 1;
            return r < ROWS && nn < a.n && ci < a.c1 && (unsigned)z < (unsigned)half && (unsigned)y < (unsigned)half;
        };
        auto issue_rows = [&](int cbase, int tid_) {
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                int s, c, hz, hy, nn, ci, z, y;
                if (row_coords(tid_ + i * NT, cbase, s, c, hz, hy, nn, ci, z, y)) {
                    const size_t si = (size_t)nn * cin + a.c0 + ci;
                    { const float4 af = a.affine[si]; xce[i] = af.x; xsc[i] = af.y; xsh[i] = af.z; }
                    const float* row = a.src1 + ((((size_t)nn * a.c1 + ci) * half + z) * half + y) * half + X0;
                    if constexpr (L == 4) {
                        const float4 t = *reinterpret_cast<const float4*>(row);
                        xraw[i][1] = t.x; xraw[i][2] = t.y; xraw[i][3] = t.z; xraw[i][4] = t.w;
                    } else {
                        const float2 t = *reinterpret_cast<const float2*>(row);
                        xraw[i][1] = t.x; xraw[i][2] = t.y;
                    }
                    if (has_l) xraw[i][0] = row[-1];
                    if (has_r) xraw[i][L + 1] = row[L];
                }
            }
        };
        auto commit_rows = [&](int cbase, int tid_, float* dstbuf) {
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int r = tid_ + i * NT;
                int s, c, hz, hy, nn, ci, z, y;
                const bool ok = row_coords(r, cbase, s, c, hz, hy, nn, ci, z, y);
                if (r < ROWS) {
                    float v[LH];
                    if (ok) {
                        const float ce = xce[i], sc = xsc[i], sh = xsh[i];
#pragma unroll
                        for (int j = 1; j <= L; ++j) v[j] = fmaf(xraw[i][j] - ce, sc, sh);
                        v[0] = has_l ? fmaf(xraw[i][0] - ce, sc, sh) : 0.f;
                        v[L + 1] = has_r ? fmaf(xraw[i][L + 1] - ce, sc, sh) : 0.f;
                    } else {
#pragma unroll
                        for (int j = 0; j < LH; ++j) v[j] = 0.f;
                    }
                    float* dst = dstbuf + (s * 8 + c) * CH1 + (hz * LH + hy) * LH;
#pragma unroll
                    for (int j = 0; j < LH; ++j) dst[j] = v[j];
                }
            }
        };

        int lane_b = lane;
        asm volatile("" : "+v"(lane_b));
        const int kq = lane_b >> 4, li = lane_b & 15;
        int aoff1[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int v = mb * 16 + li;
            int X, Y, Z, s;
            up_lattice<TE>(v, s, Z, Y, X);
            aoff1[mb] = (s * 8 + kq) * CH1 + ((Z + pz) * LH + (Y + py)) * LH + (X + px);
        }
        // This wave's weights: image rows [chunk][parity = wave][tap][k8] of cout16 floats.  They stream through a
        // wave-private, double-buffered LDS slab one QUARTER chunk (2 taps x 8 channels = 16 rows x NCO) at a time by DMA:
        // no VGPRs, and -- unlike register prefetches -- no s_waitcnt vmcnt inside the MFMA steps (vmcnt retires in order,
        // so a register prefetch waited on every step would also wait for the slow halo-row loads issued before it).
        // Slab row r = (tap&1)*8 + k8 holds cout column col at float (col + 16*(r&3)) % NCO: the 4 k rows of a B read sit
        // on different banks.
        constexpr int QF = 16 * NCO;                                    // floats per quarter slab
        constexpr int QP = QF / 256;                                    // 1-KiB DMA pieces per quarter
        float* wslab = smem + 2 * T::XS1 + wave * (2 * QF);
        const float* wimg = a.wp + (size_t)27 * a.c0_4 * a.cout16 + (size_t)wave * 64 * a.cout16;
        const int nchunk = a.c1_8 >> 3;
        auto dma_quarter = [&](int Q, int lane_) {                      // Q = chunk*4 + quarter
            float* dst = wslab + (Q & 1) * QF;
            const float* src = wimg + ((size_t)(Q >> 2) * 512 + (size_t)(Q & 3) * 16) * a.cout16;
#pragma unroll
            for (int p = 0; p < QP; ++p) {
                const int idx = p * 64 + lane_;                         // float4 index inside the slab
                const int r = idx / (NCO / 4), slot = (idx % (NCO / 4)) * 4;
                const int col = (slot + NCO - (16 * (r & 3)) % NCO) % NCO;
                int co = cob + col;
                if (co >= a.cout16) co = col % a.cout16;                // block wider than the image: masked at the store
                __builtin_amdgcn_global_load_lds((rf_gptr)(src + (size_t)r * a.cout16 + co), (rf_lptr)(dst + p * 256), 16, 0, 0);
            }
        };
        int boff1[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) boff1[nb] = kq * NCO + ((nb * 16 + li + 16 * kq) % NCO);

        dma_quarter(0, lane_b);
        issue_rows(0, tid);
        commit_rows(0, tid, xs);
        __syncthreads();                                                // rows of chunk 0 and quarter 0 have landed
        int buf = 0;
        for (int ch = 0; ch < nchunk; ++ch) {
            const bool more = ch + 1 < nchunk;
            const float* xb = xs + buf * T::XS1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int lane_o = lane, tid_o = tid;                         // opaque: index math stays inside the loop
                asm volatile("" : "+v"(lane_o), "+v"(tid_o));
                // quarter Q = 4*ch + q landed?  (wave-private slab: only this wave's own DMA, no barrier)
                // (also for Q = 0: a wave without halo rows to load has nothing else that would make it wait for its DMA)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (q < 3 || more) dma_quarter(ch * 4 + q + 1, lane_o); // into the buffer quarter Q-1 was read from
                if (q == 0 && more) issue_rows((ch + 1) * 8, tid_o);    // a whole quarter of MFMAs to land before the next wait
                const float* ws = wslab + (q & 1) * QF;
                float av[2][MB], bv[2][NB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) av[0][mb] = xb[aoff1[mb] + (((2 * q) >> 2) * LH + (((2 * q) >> 1) & 1)) * LH];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bv[0][nb] = ws[boff1[nb]];
#pragma unroll
                for (int st = 0; st < 4; ++st) {                        // step = (tap 2q + st/2, k-step st&1)
                    const int cur = st & 1, nxt = cur ^ 1;
                    const int tz = (2 * q + (st >> 1)) >> 2;             // low-res z tap of this step (compile-time)
                    auto mfma_mb = [&](int mb) {
                        // low-res row Z-1 of the first lattice plane / Z+1 of the last one is zero padding of the volume
                        if ((tz == 0 && ((LO >> mb) & 1u)) || (tz == 1 && ((HI >> mb) & 1u))) return;              // compile-time
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][mb], bv[cur][nb], acc[mb][nb], 0, 0, 0);
                    };
                    __builtin_amdgcn_sched_barrier(0);                   // same pipeline as phase A
                    mfma_mb(1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (st + 1 < 4) {
                        const int t1 = 2 * q + ((st + 1) >> 1), k1 = (st + 1) & 1;
                        const int toff = ((t1 >> 2) * LH + ((t1 >> 1) & 1)) * LH + (t1 & 1) + k1 * 4 * CH1;
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) av[nxt][mb] = xb[aoff1[mb] + toff];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) bv[nxt][nb] = ws[boff1[nb] + (st + 1) * 4 * NCO];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
                        if (mb != 1) mfma_mb(mb);
                }
            }
            if (more) {
                int tid_c = tid;
                asm volatile("" : "+v"(tid_c));
                commit_rows((ch + 1) * 8, tid_c, xs + (buf ^ 1) * T::XS1);
            }
            __syncthreads();
            buf ^= 1;
        }
    }

    // ===================================================================================================== epilogue
    // accumulators -> LDS [16 cout][P + 1] in memory order of the box -> ReLU'd float4 rows; statistics on the way
    const int lane_e = rf_lane();
    const int tid_e = wave * 64 + lane_e;
    const int kq = lane_e >> 4, li = lane_e & 15;
    float* eb = smem;
    double* red = reinterpret_cast<double*>(smem);                       // [8 waves][SPW][16 cout][2], used after the stores
    const size_t vol = (size_t)edge * edge * edge;
    constexpr int TE3 = TE * TE * TE;
#pragma unroll
    for (int nb = 0; nb < NBF; ++nb) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int v = mb * 16 + kq * 4 + r;                      // D rows of this lane: voxels 4*kq + r of the m-block
                int X, Y, Z, s;
            up_lattice<TE>(v, s, Z, Y, X);
                const int lin = s * TE3 + ((2 * Z + pz) * TE + (2 * Y + py)) * TE + (2 * X + px);
                eb[li * (P + 1) + lin] = fmaxf(acc[mb][nb][r], 0.f);
            }
        }
        __syncthreads();
        for (int q = tid_e; q < 16 * (P / 4); q += NT) {
            const int col = q / (P / 4), lin = (q % (P / 4)) * 4;
            const int co = cob + nb * 16 + col;
            const int s = lin / TE3, rem = lin % TE3;
            const int nn = n0 + s;
            if (co < a.cout && nn < a.n) {
                const float* e = eb + col * (P + 1) + lin;
                const float4 o = make_float4(e[0], e[1], e[2], e[3]);
                const int x = rem % TE, y = (rem / TE) % TE, z = rem / (TE * TE);
                *reinterpret_cast<float4*>(a.out + ((size_t)nn * a.cout + co) * vol + ((size_t)(z0 + z) * edge + (y0 + y)) * edge + (x0 + x)) = o;
            }
        }
        __syncthreads();

        if (a.stats) {
            // per (sample, cout) sum / sum of squares of the ReLU'd tile: registers -> lane groups -> waves (fixed order)
            if (SPW == 1) {
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
                if (lane_e < 16) { red[(wave * 16 + lane_e) * 2] = sm; red[(wave * 16 + lane_e) * 2 + 1] = sq; }
            } else {
                // T = 4, four samples: voxel v = mb*16 + 4*kq + r belongs to sample kq (both m-blocks = lattice planes)
                double sm = 0.0, sq = 0.0;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double v = (double)fmaxf(acc[mb][nb][r], 0.f);
                        sm += v; sq += v * v;
                    }
                red[((wave * SPW + kq) * 16 + li) * 2] = sm;
                red[((wave * SPW + kq) * 16 + li) * 2 + 1] = sq;
            }
            __syncthreads();
            if (tid_e < SPW * 16) {
                const int s = tid_e / 16, col = tid_e % 16;
                const int co = cob + nb * 16 + col, nn = n0 + s;
                if (co < a.cout && nn < a.n) {
                    double sm = 0.0, sq = 0.0;
#pragma unroll
                    for (int w = 0; w < 8; ++w) {
                        sm += red[((w * SPW + s) * 16 + col) * 2];
                        sq += red[((w * SPW + s) * 16 + col) * 2 + 1];
                    }
                    a.stats[((size_t)nn * a.cout + co) * a.stats_tiles + tile] = make_double2(sm, sq);
                }
            }
            __syncthreads();
        }
    }
    if constexpr (HALF) {
        // couts 48..55: acc4[h][i] of lane 4b + j = out(voxel 4b + i of this wave's lattice, cout 48 + 4h + j)
        const int j4 = lane_e & 3, b4i = lane_e >> 2;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = b4i * 4 + i;
                int X, Y, Z, s;
                up_lattice<TE>(v, s, Z, Y, X);
                const int lin = ((2 * Z + pz) * TE + (2 * Y + py)) * TE + (2 * X + px);
                eb[(4 * h + j4) * (P + 1) + lin] = fmaxf(acc4[h][i], 0.f);
            }
        __syncthreads();
        for (int q = tid_e; q < 8 * (P / 4); q += NT) {
            const int col = q / (P / 4), lin = (q % (P / 4)) * 4;
            const int co = cob + 48 + col;
            if (co < a.cout && n0 < a.n) {
                const float* e = eb + col * (P + 1) + lin;
                const float4 o = make_float4(e[0], e[1], e[2], e[3]);
                const int x = lin % TE, y = (lin / TE) % TE, z = lin / (TE * TE);
                *reinterpret_cast<float4*>(a.out + ((size_t)n0 * a.cout + co) * vol + ((size_t)(z0 + z) * edge + (y0 + y)) * edge + (x0 + x)) = o;
            }
        }
        __syncthreads();
        if (a.stats) {
            // registers (4 voxels) -> the 16 lanes that share lane % 4 -> waves, fixed order
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double sm = 0.0, sq = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double v = (double)fmaxf(acc4[h][i], 0.f);
                    sm += v; sq += v * v;
                }
#pragma unroll
                for (int m = 4; m < 64; m <<= 1) { sm += __shfl_xor(sm, m, 64); sq += __shfl_xor(sq, m, 64); }
                if (lane_e < 4) { red[(wave * 8 + 4 * h + lane_e) * 2] = sm; red[(wave * 8 + 4 * h + lane_e) * 2 + 1] = sq; }
            }
            __syncthreads();
            if (tid_e < 8) {
                const int co = cob + 48 + tid_e;
                if (co < a.cout && n0 < a.n) {
                    double sm = 0.0, sq = 0.0;
#pragma unroll
                    for (int w = 0; w < 8; ++w) { sm += red[(w * 8 + tid_e) * 2]; sq += red[(w * 8 + tid_e) * 2 + 1]; }
                    a.stats[((size_t)n0 * a.cout + co) * a.stats_tiles + tile] = make_double2(sm, sq);
                }
            }
            __syncthreads();
        }
    }
    };
    using Zc = std::integral_constant<unsigned, 0u>;
    constexpr bool ZSKIP = true;
    if (ZSKIP && pz == 0 && z0 == 0) run(std::integral_constant<unsigned, ZSKIP ? 0x1u : 0u>{}, Zc{});
    else if (ZSKIP && pz == 1 && z0 + TE == edge) run(Zc{}, std::integral_constant<unsigned, ZSKIP ? (1u << (MB - 1)) : 0u>{});
    else run(Zc{}, Zc{});
}

template <int TE, int SPW, int MB, int NB, bool HALF = false>
static int launch_up(const UpArgs& a, hipStream_t stream) {
    using T = UpTile<TE, SPW, MB, NB>;
    auto kern = k_conv3_up<TE, SPW, MB, NB, HALF>;
    if (T::LDS_BYTES > 65536) {
        static RfLdsOptIn opt_in;
        if (int rc = opt_in.ensure(reinterpret_cast<const void*>(kern), (int)T::LDS_BYTES, "rf_conv3d_up_k3_gn_relu")) return rc;
    }
    const unsigned gx = SPW == 1 ? (unsigned)a.n * (a.edge / TE) * (a.edge / TE) * (a.edge / TE) : (unsigned)((a.n + SPW - 1) / SPW);
    const unsigned gy = (unsigned)((a.cout16 + T::NCO - 1) / T::NCO);
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(T::NT), T::LDS_BYTES, stream, a);
    RF_CHECK_LAUNCH("rf_conv3d_up_k3_gn_relu");
    return RF_OK;
}

template <int TE, int SPW, int MB>
static int dispatch_up(const UpArgs& a, hipStream_t stream) {
    if (a.cout16 <= 16) return launch_up<TE, SPW, MB, 1>(a, stream);
    if (a.cout16 <= 32) return launch_up<TE, SPW, MB, 2>(a, stream);
    if constexpr (TE == 8) {
        if (a.cout16 == 64 && a.cout <= 56) return launch_up<TE, SPW, MB, 4, true>(a, stream);   // 3 n-blocks + 8 couts on 4x4x1 MFMAs
    }
    return launch_up<TE, SPW, MB, 4>(a, stream);
}

// Shapes the parity-split kernel takes: a low-res source, edge 4 (four samples per workgroup) or a multiple of 8, and
// enough boxes to give the 256 CUs work (small launches stay on rf_conv3d_k3_gn_relu's 128-voxel tiles).
bool rf_conv3_small_up_takes(int c0, int c1, int n, int edge, int cout);                    // conv3d_small.hip
int rf_conv3_small_up_launch(const float* src0, int c0, const float* src1, int c1, int n, const float* gn_affine,
                             const float* w_up_packed, int cout, float* out, double* stats, void* stream);

extern "C" int rf_conv3d_up_supported(int c0, int c1, int n, int edge, int cout) {
    if (c1 <= 0 || c0 < 0 || n <= 0 || cout <= 0 || !rf_is_pow2(edge) || edge < 4 || edge > 128) return 0;
    const long long gy = (rf_round_up(cout, 16) + 63) / 64;
    const long long boxes = edge == 4 ? (n + 3) / 4 : (long long)n * (edge / 8) * (edge / 8) * (edge / 8);
    return boxes * gy >= 256;
}

// which kernel rf_conv3d_up_k3_gn_relu launches for a shape: 0 = parity-split boxes (this file), 1 = position-major 4^3
// (conv3d_small.hip) -- for reporting (issued multiply-adds differ)
extern "C" int rf_conv3d_up_variant(int c0, int c1, int n, int edge, int cout) {
    if (edge == 4 && rf_conv3_small_up_takes(c0, c1, n, edge, cout)) return 1;
    return 0;
}

extern "C" int rf_conv3d_up_stats_tiles(int c0, int c1, int n, int edge, int cout) {
    return edge == 4 ? 1 : (edge / 8) * (edge / 8) * (edge / 8);
}

extern "C" int rf_conv3d_up_k3_gn_relu(const float* src0, int c0, const float* src1, int c1, int n, int edge, const float* gn_affine, const float* w_packed, int cout, float* out, double* stats, void* stream) {
    RF_REQUIRE(rf_conv3d_up_supported(c0, c1, n, edge, cout) || (c1 > 0 && c0 >= 0 && n > 0 && cout > 0 && rf_is_pow2(edge) && edge >= 4 && edge <= 128),
               RF_E_UNSUPPORTED, "rf_conv3d_up_k3_gn_relu: needs a low-res source and a power-of-two edge in 4..128 (got c1=%d edge=%d)", c1, edge);
    RF_REQUIRE((c0 == 0 || src0) && src1 && gn_affine && w_packed && out, RF_E_INVALID, "rf_conv3d_up_k3_gn_relu: null pointer");
    UpArgs a;
    a.src0 = src0; a.src1 = src1; a.affine = reinterpret_cast<const float4*>(gn_affine); a.wp = w_packed; a.out = out;
    a.c0 = c0; a.c1 = c1; a.n = n; a.edge = edge; a.cout = cout;
    a.c0_4 = rf_round_up(c0, 4); a.c1_8 = rf_round_up(c1, 8); a.cout16 = rf_round_up(cout, 16);
    a.stats = reinterpret_cast<double2*>(stats);
    a.stats_tiles = stats ? rf_conv3d_up_stats_tiles(c0, c1, n, edge, cout) : 0;
    hipStream_t s = (hipStream_t)stream;
    // whole 4^3 volumes with enough samples: position-major form (conv3d_small.hip), every zero-padding tap left out
    if (edge == 4 && rf_conv3_small_up_takes(c0, c1, n, edge, cout))
        return rf_conv3_small_up_launch(src0, c0, src1, c1, n, gn_affine, w_packed, cout, out, stats, stream);
    return edge == 4 ? dispatch_up<4, 4, 2>(a, s) : dispatch_up<8, 1, 4>(a, s);
}
