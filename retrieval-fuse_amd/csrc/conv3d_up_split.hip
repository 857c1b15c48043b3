// rf_conv3d_up_split_k3_gn_relu: the decoder form of SingleConv 'gcr' (reference model/unet.py:19-76 with the nearest x2 upsample +
// concat of :297-308 / :354-360) on whole 8^3 samples -- the same arithmetic contract as rf_conv3d_up_k3_gn_relu (conv3d_up.hip:
// upsampled channels convolved in LOW resolution with 8 pre-summed taps per output parity), evaluated on the F16 matrix cores by
// OPERAND SPLITTING instead of on v_mfma_f32_16x16x4_f32:
//
//     every fp32 operand x is carried as two f16 numbers   h = f16(x),  l = f16((x - h) * 2^11)       (x - h is exact in fp32)
//     so that x = h + l / 2^11 up to 2^-22 |x|, and        a * b  ~  ah * bh  +  (ah * bl + al * bh) / 2^11  (al * bl: 2^-22 relative, dropped).
//     An f16 x f16 product is exact in fp32.  v_mfma_f32_16x16x32_f16 accumulates in fp32; the ah*bh sums and the cross sums go to
//     SEPARATE accumulators (hi, lo) and meet once, in the epilogue:  out = hi + lo / 2^11.
//
// Three f16 MFMAs (16 cycles each for 16x16x32) replace eight fp32 MFMAs (32 cycles each for 16x16x4): 5.3x the multiply-add rate of the
// fp32 matrix path at -- measured, tools/micro/split_probe.hip -- HALF its rounding error against float64 (K = 216 ... 5184: rms 1.2e-8
// vs 2.5e-8 of sum|a b|, max 9.7e-8 vs 3.1e-7): the fp32 MFMA is a sequential fmaf chain with one rounding per product, the f16 MFMA
// rounds once per 32 products, and the 2^-22 representation error of the operands is random per element and does not accumulate.
// Activations are scaled by 2^-4 and weights by 2^4 before the split (exact): f16 overflows at 65504, so a GroupNorm output would have to
// exceed 1e6 to saturate (it is clamped, never inf), while small values lose nothing (whatever h drops, l carries).
//
// Work split (as conv3d_up.hip): one workgroup of 8 waves per sample, WAVE w OWNS OUTPUT PARITY w = (pz,py,px): its 4 m-blocks are the
// four z planes of that parity's 4^3 lattice (m-block row r = (Y,X) = (r >> 2, r & 3)), all couts (NB n-blocks of 16) -> 16 output tiles,
// 2 accumulator sets.  K order: an MFMA k-step is 4 TAPS x 8 CHANNELS -- lane group g = lane >> 4 supplies tap 4s + g, its 8 halves are 8
// consecutive channels of one voxel -- so the LDS image of the input is [8-channel group][voxel of the halo box] in 16-byte slots and every
// A operand is ONE ds_read_b128 at (voxel + tap offset):
//   A) skip channels: per 8-channel chunk 7 k-steps (27 taps + one zero-weight dummy) on the full-res halo box [10][10][10], double
//      buffered: the next chunk is staged by waves 0..3 (two voxels per thread: requested at k-steps 0 / 3, normalised + split + written to the
//      other buffer after k-steps 2 / 6) -- the SIMD arbiter favours the older wave of each pair (w, w + 4), which would otherwise idle at the chunk
//      barrier while its partner still had a conversion in front of it; one barrier per chunk;
//   B) upsampled channels: per 8-channel chunk 2 k-steps (tz = 0 / 1, lane group = (ty,tx)) on the low-res halo box [6][6][6] of the
//      chunk, all chunks staged once at kernel start (wave = channel group).
// B operands (weights) are stored in HBM in fragment order (rf_conv3_up_split_pack_weight: [k-step][n-block][h|l][lane][8 halves], the
// phase-B part once per parity), stay L2-resident (1.25 MB for 32+64->56) and go global -> VGPR with one 16-byte load per fragment, one
// k-step ahead of their MFMAs (register double buffer).  A operands are read just in time per m-block (two m-blocks of registers).
// The z-border MFMAs (output plane 0 through the dz = -1 taps, plane 7 through dz = +1: zero padding only) are not issued; the K loops are
// instantiated per pz for that.
// Epilogue: accumulators -> ReLU -> LDS tile [cout][8^3] (rows of 517 floats: conflict-free scalar writes) -> 256-byte row stores; GroupNorm
// statistics of the output (float64, fixed order) for the next layer -- or (rf_conv3d_up_split_presplit) the next layer's GroupNorm applied on
// the spot and the output written pre-split.
#include "common.h"
#include <type_traits>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int US_SY = 12, US_SZ = 120, US_ASLOTS = 1208;        // full-res halo box, slot(z,y,x) = z*120 + y*12 + x (x, y, z in 0..9)
constexpr int US_BY = 6, US_BZ = 36, US_BSLOTS = 216;           // low-res halo box, slot(Z,Y,X) = Z*36 + Y*6 + X
constexpr int US_A_PLANE = US_ASLOTS * 16;                       // bytes of one (h or l) plane of a phase-A buffer
constexpr int US_A_BUF = 2 * US_A_PLANE;
constexpr int US_B_OFF = 2 * US_A_BUF;
constexpr int US_B_PLANE = US_BSLOTS * 16;
constexpr int US_MAX_CGB = 8;                                    // low-res channel groups that fit (c1 <= 64)
constexpr int US_LDS_BYTES = US_B_OFF + US_MAX_CGB * 2 * US_B_PLANE;      // 132,608
constexpr int US_PRE_STATS = US_LDS_BYTES, US_PRE_TRIPLES = US_PRE_STATS + 64 * 16;     // pre-split epilogue: per-channel (sum, sum of squares), then triples
constexpr int US_LDS_ALLOC = US_PRE_TRIPLES + 64 * 16;                                     // 134,656
constexpr int US_T_STRIDE = 517;                                 // epilogue tile row (floats) of the whole-sample kernel: odd (bank-conflict-free scalar writes)
constexpr float US_ACT_SCALE = 1.0f / 16, US_W_SCALE = 16.0f, US_LO = 2048.0f;
constexpr bool US_ZSKIP = true, US_S4_WIDE = true;
static_assert(64 * US_T_STRIDE * 4 <= US_LDS_BYTES, "epilogue tile must fit");
}   // namespace

// ------------------------------------------------------------------------------------------------------------ weight image
// in 16-byte units (8 halves): A region [c0/8][7 steps][NB][h|l][64 lanes]  ++  B region [8 parities][c1/8][2 steps][NB][h|l][64]  ++ one
// step of zeros (the last step's prefetch lands there).  Lane l of a fragment: cout = nb*16 + (l & 15), tap group g = l >> 4.
static inline size_t us_steps(int c0, int c1) { return (size_t)(c0 / 8) * 7 + (size_t)8 * (c1 / 8) * 2 + 1; }

extern "C" size_t rf_conv3_up_split_packed_bytes(int cout, int c0, int c1) {
    return us_steps(c0, c1) * (size_t)(rf_round_up(cout, 16) / 16) * 2 * 64 * 16;
}

__global__ void k_conv3_up_split_pack(const float* __restrict__ w, int cout, int c0, int c1, int nb_count, h8* __restrict__ wp, size_t total) {
    const int cin = c0 + c1;
    const size_t nA = (size_t)(c0 / 8) * 7 * nb_count * 128;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), piece = (int)((i >> 6) & 1);
        const size_t f = i >> 7;
        const int nb = (int)(f % nb_count);
        const size_t st = f / nb_count;
        const int co = nb * 16 + (lane & 15), g = lane >> 4;
        h8 out;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double v = 0.0;
            if (i < nA) {
                const int s = (int)(st % 7), ca = (int)(st / 7), tap = 4 * s + g, ci = ca * 8 + j;
                if (tap < 27 && co < cout) v = (double)w[((size_t)co * cin + ci) * 27 + tap];
            } else {
                const size_t sb = st - (size_t)(c0 / 8) * 7;
                const int nB = c1 / 8;
                if (sb < (size_t)8 * nB * 2 && co < cout) {
                    const int s = (int)(sb % 2), cb = (int)((sb / 2) % nB), ph = (int)(sb / ((size_t)2 * nB));
                    const int ci = c0 + cb * 8 + j;
                    // per axis: parity 0: low-res tap 0 <- {0}, tap 1 <- {1,2};  parity 1: tap 0 <- {0,1}, tap 1 <- {2}   (conv3d_up.hip)
                    const int pz = ph >> 2, py = (ph >> 1) & 1, px = ph & 1, tz = s, ty = g >> 1, tx = g & 1;
                    const int z_lo = tz == 0 ? 0 : (pz ? 2 : 1), z_hi = tz == 0 ? (pz ? 1 : 0) : 2;
                    const int y_lo = ty == 0 ? 0 : (py ? 2 : 1), y_hi = ty == 0 ? (py ? 1 : 0) : 2;
                    const int x_lo = tx == 0 ? 0 : (px ? 2 : 1), x_hi = tx == 0 ? (px ? 1 : 0) : 2;
                    const float* wk = w + ((size_t)co * cin + ci) * 27;
                    for (int dz = z_lo; dz <= z_hi; ++dz)
                        for (int dy = y_lo; dy <= y_hi; ++dy)
                            for (int dx = x_lo; dx <= x_hi; ++dx) v += (double)wk[(dz * 3 + dy) * 3 + dx];      // float64 sum, split from it
                }
            }
            v *= (double)US_W_SCALE;
            v = v > 65504.0 ? 65504.0 : (v < -65504.0 ? -65504.0 : v);
            const _Float16 h = (_Float16)(float)v;
            out[j] = piece == 0 ? h : (_Float16)(float)((v - (double)(float)h) * (double)US_LO);
        }
        wp[i] = out;
    }
}

extern "C" int rf_conv3_up_split_pack_weight(const float* w_oidhw, int cout, int c0, int c1, void* w_packed, void* stream) {
    RF_REQUIRE(w_oidhw && w_packed && cout > 0 && c0 >= 0 && c1 > 0 && c0 % 8 == 0 && c1 % 8 == 0, RF_E_INVALID,
               "rf_conv3_up_split_pack_weight: needs c0 and c1 in multiples of 8 (got %d, %d)", c0, c1);
    const size_t total = rf_conv3_up_split_packed_bytes(cout, c0, c1) / 16;
    const size_t want = (total + 255) / 256;
    hipLaunchKernelGGL(k_conv3_up_split_pack, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream, w_oidhw, cout, c0, c1,
                       rf_round_up(cout, 16) / 16, reinterpret_cast<h8*>(w_packed), total);
    RF_CHECK_LAUNCH("rf_conv3_up_split_pack_weight");
    return RF_OK;
}

// ------------------------------------------------------------------------------------------------------------------- kernel
struct UpSplitArgs {
    const float* src0;
    const float* src1;
    const float4* affine;   // GroupNorm per (sample, input channel): (center, scale, shift, -) -> y = (x - center) * scale + shift
    const h8* wp;
    float* out;
    double2* stats;         // optional [n][cout][1]
    int c0, c1, n, cout;
    // pre-split output (whole-sample kernel only): the NEXT layer's GroupNorm (its gamma / beta / groups / eps over THIS layer's cout
    // channels) is applied to the ReLU'd output in the epilogue -- the workgroup holds the whole sample, so it has the statistics -- and the result is
    // written as that layer's pre-split input [n][cout/8][h | l][voxel][8 halves]; `out` is then not written
    h8* pre_out;
    const float* ngamma;
    const float* nbeta;
    int ngroups;
    float neps;
    // box kernel only: `out` is written channel-interleaved, [n][cout / 8][voxel][8 channels] fp32 (rf_conv3d_up_split_k3_gn_relu_ch8)
    int out_ch8;
    // persistent whole-sample kernel only: the voxel slots of `pre_out` in PARITY-MAJOR order -- slot index ((z & 1) 4 + (y & 1) 2 + (x & 1)) 64 + (z >> 1) 16 + (y >> 1) 4
    // + (x >> 1) instead of z 64 + y 8 + x -- so that a wave, which owns one output parity, stores 256-byte runs (RF_PRESPLIT_PARITY_MAJOR)
    int pre_pm;
};

// 8 normalised channel values of one voxel -> the two f16 pieces (scaled by 2^-4; saturating, never inf)
__device__ __forceinline__ void us_split8(const float (&y)[8], h8& h, h8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = __builtin_amdgcn_fmed3f(y[j] * US_ACT_SCALE, -65504.f, 65504.f);
        const _Float16 hh = (_Float16)v;
        h[j] = hh;
        l[j] = (_Float16)fmaf(-US_LO, (float)hh, v * US_LO);          // (v - h) * 2^11: exact either way, one v_fma_mix instead of cvt + sub + mul
    }
}

// ... of values that already carry the 2^-4 (folded into the GroupNorm triple: (x - c) (s / 16) + b / 16 rounds like ((x - c) s + b) / 16)
__device__ __forceinline__ void us_split8_scaled(const float (&y)[8], h8& h, h8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = __builtin_amdgcn_fmed3f(y[j], -65504.f, 65504.f);
        const _Float16 hh = (_Float16)v;
        h[j] = hh;
        l[j] = (_Float16)fmaf(-US_LO, (float)hh, v * US_LO);
    }
}

template <int NB>
__device__ __forceinline__ void us_mfma_block(f32x4 (&hi)[NB], f32x4 (&lo)[NB], const h8& ah, const h8& al, const h8 (&bh)[NB], const h8 (&bl)[NB]) {
    // three passes over the n-blocks: consecutive MFMAs never share an accumulator
#pragma unroll
    for (int n = 0; n < NB; ++n) hi[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[n], hi[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NB; ++n) lo[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[n], lo[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NB; ++n) lo[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[n], lo[n], 0, 0, 0);
}

template <int NB>
__global__ __launch_bounds__(512, 2) void k_conv3_up_split(UpSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pz = wave >> 2, py = (wave >> 1) & 1, px = wave & 1;
    const int n = blockIdx.x;
    const int c0 = a.c0, c1 = a.c1, cin = c0 + c1, nA = c0 >> 3, nB = c1 >> 3;
    const float4* __restrict__ aff = a.affine + (size_t)n * cin;

    // ---- the first voxels (this wave's low-res channel group, the first skip chunk) are requested before the LDS is zeroed: their HBM
    // latency passes under the zero-fill
    const float* __restrict__ sb0 = a.src0 + (size_t)n * c0 * 512;     // uniform: the voxel loads are SGPR base + a 32-bit lane offset
    float xl[8], x0[8];
    {
        const int cg = wave < nB ? wave : nB - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) xl[j] = a.src1[((size_t)n * c1 + cg * 8 + j) * 64 + lane];
        if (nA > 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x0[j] = sb0[j * 512 + tid];
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- zero the halo boxes (the padding slots are never written again)
    for (int i = tid; i < US_LDS_BYTES / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();

    // ---- stage: all low-res channel groups (wave = group, lane = low-res voxel), the first skip chunk (thread = voxel)
    for (int cg = wave; cg < nB; cg += 8) {
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 af = aff[c0 + cg * 8 + j];
            y[j] = fmaf((cg == wave ? xl[j] : a.src1[((size_t)n * c1 + cg * 8 + j) * 64 + lane]) - af.x, af.y, af.z);
        }
        h8 h, l;
        us_split8(y, h, l);
        const int slot = ((lane >> 4) + 1) * US_BZ + (((lane >> 2) & 3) + 1) * US_BY + (lane & 3) + 1;
        unsigned char* p = lds + US_B_OFF + cg * 2 * US_B_PLANE + slot * 16;
        *reinterpret_cast<h8*>(p) = h;
        *reinterpret_cast<h8*>(p + US_B_PLANE) = l;
    }
    const int vslot = ((tid >> 6) + 1) * US_SZ + (((tid >> 3) & 7) + 1) * US_SY + (tid & 7) + 1;     // this thread's voxel in the full-res halo box
    auto stage_store = [&](const float (&x)[8], int ca) {           // ca: chunk slot; past the last chunk: harmless re-staging of the last one
        float y[8];
        const int cc = ca < nA ? ca : nA - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 af = aff[cc * 8 + j];
            y[j] = fmaf(x[j] - af.x, af.y, af.z);
        }
        h8 h, l;
        us_split8(y, h, l);
        unsigned char* p = lds + (ca & 1) * US_A_BUF + vslot * 16;
        *reinterpret_cast<h8*>(p) = h;
        *reinterpret_cast<h8*>(p + US_A_PLANE) = l;
    };
    if (nA > 0) stage_store(x0, 0);

    // ---- per-lane operand addressing
    const int g = lane >> 4, rj = (lane >> 2) & 3, ri = lane & 3;
    const int abase = ((pz + 1) * US_SZ + (2 * rj + py + 1) * US_SY + (2 * ri + px + 1)) * 16;      // bytes; + 2 m SZ per m-block, + tap offset
    int atap[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int t = 4 * s + g < 27 ? 4 * s + g : 26;                                                 // the dummy tap (zero weights) reads tap 26's voxel
        atap[s] = ((t / 9 - 1) * US_SZ + ((t / 3) % 3 - 1) * US_SY + (t % 3 - 1)) * 16;
    }
    const int bbase = (pz * US_BZ + (rj + py + (g >> 1)) * US_BY + (ri + px + (g & 1))) * 16;        // + (m + tz) BZ

    // Z-BORDER TAPS: the wave's m-block 0 (pz = 0: output plane z = 0) reads nothing but zero padding through the dz = -1 taps (k-steps 0, 1 of a
    // phase-A chunk, step tz = 0 of a phase-B chunk), its m-block 3 (pz = 1: plane z = 7) through the dz = +1 taps (k-steps 5, 6; tz = 1): those
    // MFMAs would add exact zeros and are not issued (2 of 28 per phase-A chunk, 1 of 8 per phase-B chunk: 9.1 % of the kernel's MFMAs).  The
    // whole K loop + epilogue is instantiated per pz (a run-time test inside the loop makes hipcc copy the accumulators at every join).
    auto run = [&](auto PZ_) {
    constexpr int PZ = decltype(PZ_)::value;
    f32x4 hi[4][NB], lo[4][NB];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { hi[m][nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; lo[m][nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    constexpr int STEP_U4 = NB * 2 * 64;                               // 16-byte fragments rows per k-step of the weight image
    const h8* __restrict__ wB = a.wp + (size_t)nA * 7 * STEP_U4 + (size_t)wave * nB * 2 * STEP_U4 + lane;      // this parity's phase-B stream
    const h8* __restrict__ wn = nA > 0 ? a.wp + lane : wB;                                                         // next k-step to fetch
    h8 b0h[NB], b0l[NB], b1h[NB], b1l[NB];
    auto load_b = [&](h8 (&bh)[NB], h8 (&bl)[NB]) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            bh[nb] = wn[(nb * 2) * 64];
            bl[nb] = wn[(nb * 2 + 1) * 64];
        }
        wn += STEP_U4;
    };
    load_b(b0h, b0l);

    h8 ah[2], al[2];
    __syncthreads();

    // one k-step: the next step's B fragments are requested first (a full step ahead of their use), `xload` (phase A, first step of a chunk)
    // requests the next chunk's raw voxels BEHIND them -- vmcnt retires in order, so the wait for the next step's weights must not have
    // the HBM loads in front of it; A operands of m-block m+1 are fetched under the MFMAs of m-block m, `pre`: the NEXT step's m-block 0.
    // The sched_barriers pin this order (hipcc otherwise sinks every load to just before its first use).
    auto kstep = [&](auto skip_m, auto has_pre, auto&& xload, const unsigned char* ap, const unsigned char* pre, int mstride, int lplane, const h8 (&bh)[NB], const h8 (&bl)[NB], h8 (&nh)[NB], h8 (&nl)[NB]) {
        load_b(nh, nl);
        xload();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m < 3) {
                ah[(m + 1) & 1] = *reinterpret_cast<const h8*>(ap + (m + 1) * mstride);
                al[(m + 1) & 1] = *reinterpret_cast<const h8*>(ap + (m + 1) * mstride + lplane);
            } else if constexpr (decltype(has_pre)::value) {
                ah[0] = *reinterpret_cast<const h8*>(pre);
                al[0] = *reinterpret_cast<const h8*>(pre + lplane);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (m != decltype(skip_m)::value) us_mfma_block<NB>(hi[m], lo[m], ah[m & 1], al[m & 1], bh, bl);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto no_x = [] {};
    using no_skip = std::integral_constant<int, -1>;
    using skip_lo = std::integral_constant<int, US_ZSKIP && PZ == 0 ? 0 : -1>;      // steps whose taps all have dz = -1 / tz = 0
    using skip_hi = std::integral_constant<int, US_ZSKIP && PZ == 1 ? 3 : -1>;      // ... dz = +1 / tz = 1

    // ---- phase A: skip channels.  No branch inside a chunk (hipcc's s_waitcnt insertion assumes the worst at every join): the last chunk
    // re-loads its own channels and stages them into the idle buffer.
    // Staging of the next chunk is the job of waves 0..3 (PZ = 0) alone, two voxels per thread (z and z + 4): the SIMD arbiter favours the older
    // wave of each pair (w, w + 4), which therefore finishes its seven k-steps ~4 k cycles before its partner and used to idle at the chunk
    // barrier, while the partner -- last to finish -- still had its own conversion in front of that barrier.  Now the early wave converts (once
    // after k-step 2, once after k-step 6, each time while its partner has the matrix pipe to itself) and the late wave goes straight to the
    // barrier.  Requests sit BEHIND the step's weight loads (vmcnt retires in order); the voxels land within two k-steps (measured 1.8-2.1 k
    // cycles), three / four pass before they are used.
    auto chunk_a = [&](int ca, h8 (&ch)[NB], h8 (&cl)[NB], h8 (&nh)[NB], h8 (&nl)[NB]) {
        float x[8];
        const bool more = ca + 1 < nA;
        const int cx = more ? ca + 1 : ca;                              // past the last chunk: harmless re-staging of the last one into the idle buffer
        const float* __restrict__ sbx = sb0 + (size_t)cx * 8 * 512;      // uniform
        auto xload_a = [&] {
            if constexpr (PZ == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = sbx[j * 512 + (tid & 255)];
            }
        };
        auto xload_b = [&] {
            if constexpr (PZ == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = sbx[j * 512 + 256 + (tid & 255)];
            }
        };
        auto convert_store = [&](int half) {
            if constexpr (PZ == 0) {
                __builtin_amdgcn_sched_barrier(0);
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 af = aff[cx * 8 + j];
                    y[j] = fmaf(x[j] - af.x, af.y, af.z);
                }
                h8 h, l;
                us_split8(y, h, l);
                unsigned char* p = lds + ((ca + 1) & 1) * US_A_BUF + (vslot + half * 4 * US_SZ) * 16;
                *reinterpret_cast<h8*>(p) = h;
                *reinterpret_cast<h8*>(p + US_A_PLANE) = l;
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        const unsigned char* buf = lds + (ca & 1) * US_A_BUF + abase;
        ah[0] = *reinterpret_cast<const h8*>(buf + atap[0]);
        al[0] = *reinterpret_cast<const h8*>(buf + atap[0] + US_A_PLANE);
        constexpr int MS = 2 * US_SZ * 16;
        kstep(skip_lo{}, std::true_type{}, xload_a, buf + atap[0], buf + atap[1], MS, US_A_PLANE, ch, cl, nh, nl);     // taps 0..3:   dz = -1
        kstep(skip_lo{}, std::true_type{}, no_x, buf + atap[1], buf + atap[2], MS, US_A_PLANE, nh, nl, ch, cl);        // taps 4..7:   dz = -1
        kstep(no_skip{}, std::true_type{}, no_x, buf + atap[2], buf + atap[3], MS, US_A_PLANE, ch, cl, nh, nl);        // taps 8..11
        convert_store(0);
        kstep(no_skip{}, std::true_type{}, xload_b, buf + atap[3], buf + atap[4], MS, US_A_PLANE, nh, nl, ch, cl);
        kstep(no_skip{}, std::true_type{}, no_x, buf + atap[4], buf + atap[5], MS, US_A_PLANE, ch, cl, nh, nl);        // taps 16..19
        kstep(skip_hi{}, std::true_type{}, no_x, buf + atap[5], buf + atap[6], MS, US_A_PLANE, nh, nl, ch, cl);        // taps 20..23: dz = +1
        wn = more ? wn : wB;                                            // the last phase-A step fetches the first phase-B step
        kstep(skip_hi{}, std::false_type{}, no_x, buf + atap[6], buf + atap[6], MS, US_A_PLANE, ch, cl, nh, nl);       // taps 24..26 + the dummy
        convert_store(1);
        __syncthreads();
    };
    for (int ca = 0; ca < nA; ca += 2) {
        chunk_a(ca, b0h, b0l, b1h, b1l);                                 // 7 steps: the fetched step ends up in b1
        if (ca + 1 < nA) chunk_a(ca + 1, b1h, b1l, b0h, b0l);
        else {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) { b0h[nb] = b1h[nb]; b0l[nb] = b1l[nb]; }
        }
    }

    // ---- phase B: upsampled channels in low resolution
    {
        const unsigned char* bb = lds + US_B_OFF + bbase;
        ah[0] = *reinterpret_cast<const h8*>(bb);
        al[0] = *reinterpret_cast<const h8*>(bb + US_B_PLANE);
        for (int cb = 0; cb < nB; ++cb) {
            const unsigned char* ap = bb + cb * 2 * US_B_PLANE;
            kstep(skip_lo{}, std::true_type{}, no_x, ap, ap + US_BZ * 16, US_BZ * 16, US_B_PLANE, b0h, b0l, b1h, b1l);                                   // tz = 0
            kstep(skip_hi{}, std::true_type{}, no_x, ap + US_BZ * 16, cb + 1 < nB ? ap + 2 * US_B_PLANE : ap, US_BZ * 16, US_B_PLANE, b1h, b1l, b0h, b0l);   // tz = 1
        }
    }
    __syncthreads();

    // ---- epilogue: out = relu(hi + lo / 2^11) -> LDS tile [cout][z][y][x] -> float4 rows
    // (what the epilogue derives from the thread index is computed here, from an opaque copy: hoisted above the K loops those values sit in
    // spilled registers through 44 k-steps -- 430 MB of scratch traffic per launch in the PMC counters)
    int te = tid;
    asm volatile("" : "+v"(te));
    float* e = reinterpret_cast<float*>(lds);
    {
        const int col = te & 15, yj = (te >> 4) & 3;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lin = (2 * m + pz) * 64 + (2 * yj + py) * 8 + 2 * r + px;
                    e[(nb * 16 + col) * US_T_STRIDE + lin] = fmaxf(fmaf(lo[m][nb][r], 1.0f / US_LO, hi[m][nb][r]), 0.f);
                }
    }
    __syncthreads();
    const int cout = a.cout;
    if (a.pre_out) {
        // ---- pre-split output: statistics of the whole sample -> the next layer's triples -> normalise, split, 16-byte slots
        double2* chst = reinterpret_cast<double2*>(lds + US_PRE_STATS);
        float4* trip = reinterpret_cast<float4*>(lds + US_PRE_TRIPLES);
        {
            const int co = te >> 3, part = te & 7;
            double sm = 0.0, sq = 0.0;
            if (co < cout) {
#pragma unroll 8
                for (int i = 0; i < 64; ++i) {
                    const float v = e[co * US_T_STRIDE + part + 8 * i];
                    sm += (double)v; sq += (double)v * v;
                }
            }
#pragma unroll
            for (int msk = 1; msk < 8; msk <<= 1) { sm += __shfl_xor(sm, msk, 64); sq += __shfl_xor(sq, msk, 64); }
            if (part == 0 && co < cout) {
                chst[co] = make_double2(sm, sq);
                if (a.stats) a.stats[(size_t)n * cout + co] = make_double2(sm, sq);
            }
        }
        __syncthreads();
        if (te < cout) {                                             // as rf_gn_from_stats: group sums in channel order, float64
            const int cpg = cout / a.ngroups, ca = (te / cpg) * cpg;
            double sm = 0.0, sq = 0.0;
            for (int c = ca; c < ca + cpg; ++c) { sm += chst[c].x; sq += chst[c].y; }
            const double count = (double)cpg * 512.0, mean = sm / count;
            double var = sq / count - mean * mean;
            if (var < 0.0) var = 0.0;
            trip[te] = gn_affine(mean, 1.0 / sqrt(var + (double)a.neps), a.ngamma[te], a.nbeta[te]);
        }
        __syncthreads();
        h8* __restrict__ po = a.pre_out + (size_t)n * (cout >> 3) * 2 * 512 + te;
        for (int sg = 0; sg < (cout >> 3); ++sg) {
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 t4 = trip[sg * 8 + j];
                y[j] = fmaf(e[(sg * 8 + j) * US_T_STRIDE + te] - t4.x, t4.y, t4.z);
            }
            h8 h, l;
            us_split8(y, h, l);
            po[(size_t)sg * 2 * 512] = h;
            po[(size_t)sg * 2 * 512 + 512] = l;
        }
        return;
    }
    float* __restrict__ o = a.out + (size_t)n * cout * 512;
    // rows of 517 floats: the scalar tile writes above (16 couts x 2 y per half wave) and these row reads hit 32 different banks; a row
    // stride that is a multiple of 4 (needed for 16-byte reads) leaves every write 4-way conflicted.  A wave stores 256 contiguous bytes.
#pragma unroll 4
    for (int co = 0; co < cout; ++co) o[(size_t)co * 512 + te] = e[co * US_T_STRIDE + te];
    if (a.stats) {
        // per cout: eight threads sum 64 values each (voxels part, part + 8, ...; float64), then the eight partial sums in a fixed order
        const int co = te >> 3, part = te & 7;
        double sm = 0.0, sq = 0.0;
        if (co < cout) {
#pragma unroll 8
            for (int i = 0; i < 64; ++i) {
                const float v = e[co * US_T_STRIDE + part + 8 * i];
                sm += (double)v; sq += (double)v * v;
            }
        }
#pragma unroll
        for (int msk = 1; msk < 8; msk <<= 1) { sm += __shfl_xor(sm, msk, 64); sq += __shfl_xor(sq, msk, 64); }
        if (part == 0 && co < cout) a.stats[(size_t)n * cout + co] = make_double2(sm, sq);
    }
    };   // run
    if (pz == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
}

// ------------------------------------------------------------------------------------------- persistent form, MFMA roles swapped
// k_conv3_up_split_pp: the same layer (whole 8^3 samples, 49..64 couts, pre-split output only) as a PERSISTENT kernel -- one workgroup per CU walks
// samples n, n + grid, ... -- built so that nothing but the epilogue of a sample stands between its last MFMA and the next sample's first:
//  * MFMA ROLES SWAPPED: the weights are the A operand (M = 16 couts), the voxels the B operand (N = 16 voxels).  An accumulator lane then holds 4 consecutive M
//    rows of ONE voxel; with the rows of a cout-block pair permuted (row 4 g' + r of block 2 p + j  <->  cout 32 p + 8 g' + 4 j + r: a per-lane gather out of the
//    SAME weight image, no second pack) lane (g, v) holds the 8 couts 32 p + 8 g .. + 7 of voxel v: exactly one 16-byte slot of the pre-split output.  ReLU,
//    statistics, the next layer's GroupNorm, the h / l split and the stores all leave from registers -- no LDS tile (the 132 KB tile of k_conv3_up_split aliases
//    every image, which is what forbade staging sample i + 1 under sample i).
//  * TILE 8 VOXEL BLOCKS x 2 COUT BLOCKS per wave: wave = (cout half ch, (py, px)); its voxel blocks are the 8 z planes of the (py, px) column, i.e. BOTH z
//    parities.  tools/pp_ablation.py on the first form (4 x 4 tile, tools/variants/): the k-loops wait for the weight stream through the L1 -- 8 KB per wave and
//    k-step, 43 of the L1's 64 B/clk at full MFMA rate.  Here a wave needs its cout half only: 4 KB per k-step in phase A (phase B: 8 KB, two parities -- every
//    (parity, cout block) fragment has exactly one owner, the minimum), a k-step's fragments are 16 registers, and a SECOND set (one whole k-step ahead, as
//    k_conv3_up_split has it) costs no more than that kernel's single set.  The z-border k-steps are skipped by every wave (plane 0 / plane 7): one code path.
//  * PHASE ORDER B -> A: the low-res images are dead once phase B is over, so the NEXT sample's low-res voxels are staged into them during phase A (chunk 1), and
//    its first skip chunk goes into the idle halo buffer during the last chunk (c0 / 8 even): the per-sample prologue (zero fill, staging, barrier: ~10 k of ~98 k
//    cycles) is paid once per workgroup.  Every thread stages one voxel per chunk.
//  * Every global access of the sample loop is a BUFFER access (descriptor in SGPRs + one 32-bit lane offset + scalar offset): with flat / global addressing
//    hipcc forms 64-bit lane addresses per (k-step, fragment) outside the sample loop and spills them; a spill reload is a scratch load in the in-order vmcnt
//    queue, i.e. a wait for every staging load in front of it, between MFMAs.  The input GroupNorm's triples come out of an LDS table (a global load of them
//    is a VECTOR load once the loop has stores: hipcc no longer proves the table unclobbered).
// Statistics of a channel: float64 per lane (8 planes), a butterfly over the 16 voxel lanes of a row, then the four (py, px) waves in order.
namespace {
constexpr int PP_CHST = US_LDS_BYTES;                            // [2 cout halves][4 (py, px)][32 couts] double2
constexpr int PP_TRIP = PP_CHST + 8 * 32 * 16;                   // [64 couts] float4
constexpr int PP_CHS = PP_TRIP + 64 * 16;                        // [64 couts] double2: a channel's sums over the sample
constexpr int PP_GB = PP_CHS + 64 * 16;                          // [64 couts] float2: the next layer's gamma, beta
constexpr int PP_AFF = PP_GB + 64 * 8;                           // [2 sample parities][center | scale | shift][128 input channels] floats: the input GroupNorm's triples, SoA
constexpr int PP_LDS_ALLOC = PP_AFF + 2 * 3 * 128 * 4;           // 142,336
}   // namespace

// one value of the lane's row partner under a DPP control (float64 as two dwords)
template <int CTRL>
__device__ __forceinline__ double pp_dpp_f64(double v) {
    const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi_, lo_);
}
// 8 values per lane, summed over the 16 lanes of a DPP row: a butterfly that halves the value count per level (partner lanes n ^ 15, n ^ 7, n ^ 2: every partner
// agrees with the lane on the bits already used), then one plain exchange with lane n ^ 1 -- lanes n and n ^ 1 end up with the row's total of value n >> 1
__device__ __forceinline__ double pp_row16_transpose_sum8(double (&d)[8], int n) {
#define PP_LEVEL(W_, CTRL_, BIT_)                                                                   \
    {                                                                                               \
        const bool up_ = (n & BIT_) != 0;                                                           \
        _Pragma("unroll") for (int i = 0; i < W_; ++i) {                                            \
            const double keep_ = up_ ? d[i + W_] : d[i], send_ = up_ ? d[i] : d[i + W_];            \
            d[i] = keep_ + pp_dpp_f64<CTRL_>(send_);                                                \
        }                                                                                           \
    }
    PP_LEVEL(4, 0x140, 8)
    PP_LEVEL(2, 0x141, 4)
    PP_LEVEL(1, 0x4E, 2)
#undef PP_LEVEL
    return d[0] + pp_dpp_f64<0xB1>(d[0]);
}

// development ablations (tools/pp_ablation.py; wrong results): bit 0 = no staging (requests, conversions), 1 = no epilogue, 2 = no weight loads, 3 = no MFMAs
#ifndef RF_PP_ABL
#define RF_PP_ABL 0
#endif
#ifdef RF_PP_STAMPS
// development build (tools/pp_stamps.py): s_memtime at the phase borders of every workgroup's 4th sample, waves 0 and 4, kept in SGPRs until the kernel's end
__device__ unsigned long long g_pp_stamps[1024 * 2 * 16];
extern "C" int rft_pp_read_stamps(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pp_stamps), sizeof(g_pp_stamps)); }
#define PP_STAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st[i] = it == 3 ? t_ : st[i]; } while (0)
#else
#define PP_STAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(512, 2) void k_conv3_up_split_pp(UpSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#ifdef RF_PP_STAMPS
    unsigned long long st[16] = {};
    int it = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ch = wave & 1, q = wave >> 1, py = q >> 1, px = q & 1;         // cout half, (y, x) parity of the wave's voxel column
    const int c0 = a.c0, c1 = a.c1, cin = c0 + c1, nA = c0 >> 3, nB = c1 >> 3;
    const int G = (int)gridDim.x;

    // ---- the workgroup's first sample: low-res groups (wave = group) and skip chunk 0 (thread = voxel), requested before the zero fill
    {
        const int n = blockIdx.x;
        const float4* __restrict__ aff = a.affine + (size_t)n * cin;
        const float* __restrict__ sb0 = a.src0 + (size_t)n * c0 * 512;
        float xl[8], x0[8];
        const int cg = wave < nB ? wave : nB - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) xl[j] = a.src1[((size_t)n * c1 + cg * 8 + j) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 8; ++j) x0[j] = sb0[j * 512 + tid];
        __builtin_amdgcn_sched_barrier(0);
        for (int i = tid; i < US_LDS_BYTES / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0u, 0u, 0u, 0u);
        if (tid < a.cout) reinterpret_cast<float2*>(lds + PP_GB)[tid] = make_float2(a.ngamma[tid], a.nbeta[tid]);
        if (tid < cin) {
            const float4 t4 = aff[tid];
            float* tb = reinterpret_cast<float*>(lds + PP_AFF);
            tb[tid] = t4.x; tb[128 + tid] = t4.y; tb[256 + tid] = t4.z;
        }
        __syncthreads();
        if (wave < nB) {
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 af = aff[c0 + cg * 8 + j];
                y[j] = fmaf(xl[j] - af.x, af.y, af.z);
            }
            h8 h, l;
            us_split8(y, h, l);
            const int slot = ((lane >> 4) + 1) * US_BZ + (((lane >> 2) & 3) + 1) * US_BY + (lane & 3) + 1;
            unsigned char* p = lds + US_B_OFF + cg * 2 * US_B_PLANE + slot * 16;
            *reinterpret_cast<h8*>(p) = h;
            *reinterpret_cast<h8*>(p + US_B_PLANE) = l;
        }
        {
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 af = aff[j];
                y[j] = fmaf(x0[j] - af.x, af.y, af.z);
            }
            h8 h, l;
            us_split8(y, h, l);
            const int vs = ((tid >> 6) + 1) * US_SZ + (((tid >> 3) & 7) + 1) * US_SY + (tid & 7) + 1;
            unsigned char* p = lds + vs * 16;
            *reinterpret_cast<h8*>(p) = h;
            *reinterpret_cast<h8*>(p + US_A_PLANE) = l;
        }
    }

    // ---- per-lane operand addressing.  Voxel operand: lane group g = tap of the k-step, lane & 15 = (Y, X) of the (py, px) lattice, voxel block vb = z plane
    const int g = lane >> 4, rj = (lane >> 2) & 3, ri = lane & 3;
    const int abase = (US_SZ + (2 * rj + py + 1) * US_SY + (2 * ri + px + 1)) * 16;            // + vb US_SZ 16 + tap offset
    // the lane group's tap offset of k-step s (tap 4 s + g; the dummy 28th tap reads tap 26's voxel), in 16-byte slots, three 10-bit fields per register
    int tpk[3] = {0, 0, 0};
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int t = 4 * s + g < 27 ? 4 * s + g : 26;
        tpk[s / 3] |= (((t / 9 - 1) * US_SZ + ((t / 3) % 3 - 1) * US_SY + (t % 3 - 1)) & 1023) << (10 * (s % 3));
    }
    auto atap = [&](int s) {
        int t = tpk[s / 3];
        asm volatile("" : "+v"(t));                                  // unpacked where it is used, not seven registers' worth outside the sample loop
        return __builtin_amdgcn_sbfe(t, 10 * (s % 3), 10) * 16;
    };
    // low-res operand of output plane z = vb (parity pz = vb & 1), k-step tz, lane group (ty, tx): halo voxel ((vb + 1 >> 1) + tz, rj + py + ty, ri + px + tx)
    const int bbase = ((rj + py + (g >> 1)) * US_BY + (ri + px + (g & 1))) * 16;
    // weight operand: row m = lane & 15 of cout-block 2 ch + j is cout 32 ch + 8 (m >> 2) + 4 j + (m & 3), which the image (fragment order [n-block][h | l][lane],
    // lane = 16 g + (cout & 15)) keeps in n-block 2 ch + (m >> 3) at lane 16 g + 8 ((m >> 2) & 1) + 4 j + (m & 3)
    const int m16 = lane & 15;
    const int wlb = (((m16 >> 3) * 128 + 8 * ((m16 >> 2) & 1) + (m16 & 3) + 16 * g) * 16);
    constexpr int STEPB = 4 * 2 * 64 * 16;                           // bytes per k-step of the image (NB = 4)
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<h8*>(a.wp), 0, (int)(((size_t)nA * 7 + 16 * (size_t)nB + 1) * STEPB), 0x00020000);
    const int wA = 4096 * ch;                                        // byte offsets of the wave's streams: phase A, phase B for pz = 0 / 1
    const int wB0 = (nA * 7 + (2 * py + px) * nB * 2) * STEPB + 4096 * ch;
    const int wB1 = (nA * 7 + (4 + 2 * py + px) * nB * 2) * STEPB + 4096 * ch;
    auto ldw = [&](h8 (&R)[4], int base) {                           // a cout half's fragments of one k-step: [j * 2 + (h | l)]
        if constexpr ((RF_PP_ABL & 4) != 0) {
#pragma unroll
            for (int f = 0; f < 4; ++f) asm volatile("" : "+v"(R[f]));
            return;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) R[j * 2 + hl] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rW, wlb, base + (64 * hl + 4 * j) * 16, 0));
    };
    h8 T0[4], T1[4], T2[4], T3[4];                                   // phase A: T0 / T1 alternate; phase B: (T0, T2) / (T1, T3) = (pz 0, pz 1) alternate
    ldw(T0, wB0);
    ldw(T2, wB1);

    f32x4 hi[8][2], lo[8][2];                                        // [z plane][cout block of the half]
    h8 vh[2], vl[2];
    auto mfma6 = [&](int vb, const h8 (&W)[4], const h8& xh, const h8& xl) {
        if constexpr ((RF_PP_ABL & 8) != 0) {
            hi[vb][0][0] += (float)W[0][0] * (float)xh[0] + (float)W[1][0] * (float)xl[0];
            hi[vb][1][0] += (float)W[2][0] * (float)xh[0] + (float)W[3][0] * (float)xl[0];
            return;
        }
        hi[vb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[0], xh, hi[vb][0], 0, 0, 0);
        hi[vb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[2], xh, hi[vb][1], 0, 0, 0);
        lo[vb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[1], xh, lo[vb][0], 0, 0, 0);
        lo[vb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[3], xh, lo[vb][1], 0, 0, 0);
        lo[vb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[0], xl, lo[vb][0], 0, 0, 0);
        lo[vb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[2], xl, lo[vb][1], 0, 0, 0);
    };
    // phase-A k-step on the 8 planes at `vp` (plane stride US_SZ slots): the NEXT k-step's fragments first (a whole k-step ahead), `xload` (staging requests) behind
    // them -- vmcnt retires in order --, plane p + 1 read under the MFMAs of plane p.  On entry (vh[0], vl[0]) hold plane 0; has_pre: `pre` = plane 0 of the next k-step.
    auto kstep_a = [&](auto skip_c, auto has_pre, auto&& xload, const unsigned char* vp, const unsigned char* pre, h8 (&C)[4], h8 (&N)[4], int wnext) {
        constexpr int skip = decltype(skip_c)::value;
        ldw(N, wnext);
        xload();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int vb = 0; vb < 8; ++vb) {
            if (vb < 7) {
                vh[(vb + 1) & 1] = *reinterpret_cast<const h8*>(vp + (vb + 1) * (US_SZ * 16));
                vl[(vb + 1) & 1] = *reinterpret_cast<const h8*>(vp + (vb + 1) * (US_SZ * 16) + US_A_PLANE);
            } else if constexpr (decltype(has_pre)::value) {
                vh[0] = *reinterpret_cast<const h8*>(pre);
                vl[0] = *reinterpret_cast<const h8*>(pre + US_A_PLANE);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (vb != skip) mfma6(vb, C, vh[vb & 1], vl[vb & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // phase-B k-step tz of one 8-channel group at `bp` (low-res halo plane stride US_BZ slots): output plane vb reads halo plane ((vb + 1) >> 1) + tz, so planes
    // P = 1..4 serve vb = {2 P - 1, 2 P} (tz = 0; vb = 0 reads padding: skipped) or {2 P - 3, 2 P - 2} (tz = 1; vb = 7 skipped); weights by the plane's z parity
    // (C0 / C1).  On entry (vh[0], vl[0]) hold plane 1; `pre` = plane 1 of the next k-step.
    auto kstep_b = [&](auto TZ_, const unsigned char* bp, const unsigned char* pre, h8 (&C0)[4], h8 (&C1)[4], h8 (&N0)[4], h8 (&N1)[4], int wn0, int wn1) {
        constexpr int TZ = decltype(TZ_)::value;
        ldw(N0, wn0);
        ldw(N1, wn1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                // halo plane P = k + 1 in ring slot k & 1
            const unsigned char* qn = k < 3 ? bp + (k + 2) * (US_BZ * 16) : pre;
            vh[(k + 1) & 1] = *reinterpret_cast<const h8*>(qn);
            vl[(k + 1) & 1] = *reinterpret_cast<const h8*>(qn + US_B_PLANE);
            __builtin_amdgcn_sched_barrier(0);
            const int vodd = 2 * (k + 1) - 1 - 2 * TZ, veven = vodd + 1;      // the pz = 1 plane and the pz = 0 plane this halo plane serves
            if (vodd >= 0 && vodd <= 7 && !(TZ == 1 && vodd == 7)) mfma6(vodd, C1, vh[k & 1], vl[k & 1]);
            if (veven >= 0 && veven <= 7 && !(TZ == 0 && veven == 0)) mfma6(veven, C0, vh[k & 1], vl[k & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // x <- (x - center) * scale + shift with the 8 triples of a channel group out of the LDS table (uniform addresses), in two stages of 8 / 16 registers: the
    // conversion sits at the kernel's register peak (accumulators + two k-steps' weights + the staged voxels)
    auto normalise8 = [&](float (&x)[8], int table, int chn) {
        const float* tb = reinterpret_cast<const float*>(lds + PP_AFF) + table * 384 + chn;
        {
            const float4 c0_ = *reinterpret_cast<const float4*>(tb), c1_ = *reinterpret_cast<const float4*>(tb + 4);
            x[0] -= c0_.x; x[1] -= c0_.y; x[2] -= c0_.z; x[3] -= c0_.w; x[4] -= c1_.x; x[5] -= c1_.y; x[6] -= c1_.z; x[7] -= c1_.w;
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const float4 s0_ = *reinterpret_cast<const float4*>(tb + 128), s1_ = *reinterpret_cast<const float4*>(tb + 132);
            const float4 h0_ = *reinterpret_cast<const float4*>(tb + 256), h1_ = *reinterpret_cast<const float4*>(tb + 260);
            x[0] = fmaf(x[0], s0_.x, h0_.x); x[1] = fmaf(x[1], s0_.y, h0_.y); x[2] = fmaf(x[2], s0_.z, h0_.z); x[3] = fmaf(x[3], s0_.w, h0_.w);
            x[4] = fmaf(x[4], s1_.x, h1_.x); x[5] = fmaf(x[5], s1_.y, h1_.y); x[6] = fmaf(x[6], s1_.z, h1_.z); x[7] = fmaf(x[7], s1_.w, h1_.w);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto no_x = [] {};
    using no_skip = std::integral_constant<int, -1>;
    using skip_lo = std::integral_constant<int, US_ZSKIP ? 0 : -1>;      // k-steps whose taps all have dz = -1: plane 0 reads padding
    using skip_hi = std::integral_constant<int, US_ZSKIP ? 7 : -1>;      // ... dz = +1: plane 7

    // 1 / (values per GroupNorm group of the output), float64, kept as two scalars (a VGPR pair across the sample loop is a spill candidate)
    int icnt_lo, icnt_hi;
    {
        const double ic = 1.0 / ((double)(a.cout / a.ngroups) * 512.0);
        icnt_lo = __builtin_amdgcn_readfirstlane(__double2loint(ic));
        icnt_hi = __builtin_amdgcn_readfirstlane(__double2hiint(ic));
    }
    __syncthreads();                                                 // the first sample's images are in place
    int sp = 0;                                                      // which triple table is this sample's (the other one is filled for the next sample in chunk 0)
    for (int n = blockIdx.x; n < a.n; n += G, sp ^= 1) {
        const int nn = n + G < a.n ? n + G : n;                      // the sample staged under this one (the last one re-stages itself: harmless)
        PP_STAMP(0);
#pragma unroll
        for (int vb = 0; vb < 8; ++vb)
#pragma unroll
            for (int j = 0; j < 2; ++j) { hi[vb][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; lo[vb][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

        // ---- phase B: upsampled channels in low resolution (the two z parities' pre-summed taps), 2 k-steps per 8-channel group
        {
            int bb_ = bbase;
            asm volatile("" : "+v"(bb_));
            const unsigned char* bb = lds + US_B_OFF + bb_ + US_BZ * 16;       // halo plane 1
            vh[0] = *reinterpret_cast<const h8*>(bb);
            vl[0] = *reinterpret_cast<const h8*>(bb + US_B_PLANE);
            int w0 = wB0 + STEPB, w1 = wB1 + STEPB;
            for (int cb = 0; cb < nB; ++cb) {
                const unsigned char* bp = bb + cb * 2 * US_B_PLANE - US_BZ * 16;       // halo plane 0 of the group
                const bool last = cb + 1 == nB;
                kstep_b(std::integral_constant<int, 0>{}, bp, bp + US_BZ * 16, T0, T2, T1, T3, w0, w1);
                w0 += STEPB; w1 += STEPB;
                // (after the last k-step: phase A's first fragments go to T0; T2 takes any fragment of the image)
                kstep_b(std::integral_constant<int, 1>{}, bp, last ? bp + US_BZ * 16 : bp + 2 * US_B_PLANE + US_BZ * 16, T1, T3, T0, T2, last ? wA : w0, last ? wA : w1);
                w0 += STEPB; w1 += STEPB;
            }
        }
        PP_STAMP(1);

        // ---- phase A: skip channels, 7 k-steps per chunk on the double-buffered halo box (chunk ca in buffer ca & 1; c0 / 8 is even).  Every thread stages one voxel
        // per chunk -- the next chunk's (after the last chunk: the NEXT sample's chunk 0), requested behind k-step 0, converted behind k-step 2 -- and in chunk 1 one
        // low-res voxel of the next sample (group = wave, requested behind k-step 3, converted behind k-step 5); chunk 0 also copies the next sample's triple table.
        auto chunk_a = [&](int ca, auto BK_, h8 (&CA)[4], h8 (&NA)[4]) {
            constexpr int BK = decltype(BK_)::value;                 // 1: low-res staging in this chunk; -2: the next sample's triple table; -1: neither
            float x[8];
            const bool more = ca + 1 < nA;
            const int cx = more ? ca + 1 : 0, nx = more ? n : nn;
            const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0) + ((size_t)nx * c0 + cx * 8) * 512, 0, 8 * 512 * 4, 0x00020000);
            const int cgb = wave < nB ? wave : nB - 1;
            float4 afn;                                              // chunk 0: this thread's entry of the NEXT sample's table
            auto xload_a = [&] {
                if constexpr (!(RF_PP_ABL & 1)) {
                    int t4_ = tid;
                    asm volatile("" : "+v"(t4_));
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rS, t4_ * 4, j * 2048, 0));
                    if constexpr (BK == -2) {
                        const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.affine) + (size_t)nn * cin, 0, cin * 16, 0x00020000);
                        afn = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rF, (t4_ < cin ? t4_ : cin - 1) * 16, 0, 0));
                    }
                }
            };
            auto xload_low = [&] {
                if constexpr (BK == 1 && !(RF_PP_ABL & 1)) {
                    const __amdgpu_buffer_rsrc_t rL = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src1) + ((size_t)nn * c1 + cgb * 8) * 64, 0, 8 * 64 * 4, 0x00020000);
                    int l4_ = tid;
                    asm volatile("" : "+v"(l4_));
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rL, (l4_ & 63) * 4, j * 256, 0));
                }
            };
            auto convert_store = [&] {
                if constexpr (!(RF_PP_ABL & 1)) {
                    __builtin_amdgcn_sched_barrier(0);
                    normalise8(x, more ? sp : sp ^ 1, cx * 8);
                    h8 h, l;
                    us_split8(x, h, l);
                    int vs = tid;
                    asm volatile("" : "+v"(vs));
                    const int tix = vs;
                    vs = ((vs >> 6) + 1) * US_SZ + (((vs >> 3) & 7) + 1) * US_SY + (vs & 7) + 1;
                    unsigned char* p = lds + ((ca + 1) & 1) * US_A_BUF + vs * 16;
                    *reinterpret_cast<h8*>(p) = h;
                    *reinterpret_cast<h8*>(p + US_A_PLANE) = l;
                    if constexpr (BK == -2) {
                        float* tb = reinterpret_cast<float*>(lds + PP_AFF) + (sp ^ 1) * 384 + (tix < cin ? tix : cin - 1);
                        tb[0] = afn.x; tb[128] = afn.y; tb[256] = afn.z;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            auto convert_low = [&] {
                if constexpr (BK == 1 && !(RF_PP_ABL & 1)) {
                    __builtin_amdgcn_sched_barrier(0);
                    normalise8(x, sp ^ 1, c0 + cgb * 8);
                    h8 h, l;
                    us_split8(x, h, l);
                    int ls = tid;
                    asm volatile("" : "+v"(ls));
                    ls = (((ls >> 4) & 3) + 1) * US_BZ + (((ls >> 2) & 3) + 1) * US_BY + (ls & 3) + 1;
                    unsigned char* p = lds + US_B_OFF + cgb * 2 * US_B_PLANE + ls * 16;
                    *reinterpret_cast<h8*>(p) = h;
                    *reinterpret_cast<h8*>(p + US_B_PLANE) = l;
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            // this chunk's halo buffer + the lane's voxel, opaque: hipcc otherwise keeps (buffer, tap) address variants across the whole sample loop
            int abuf = abase + (ca & 1) * US_A_BUF;
            asm volatile("" : "+v"(abuf));
            const unsigned char* buf = lds + abuf;
            vh[0] = *reinterpret_cast<const h8*>(buf + atap(0));
            vl[0] = *reinterpret_cast<const h8*>(buf + atap(0) + US_A_PLANE);
            const int wn = wA + (ca * 7 + 1) * STEPB;
            kstep_a(skip_lo{}, std::true_type{}, xload_a, buf + atap(0), buf + atap(1), CA, NA, wn);
            kstep_a(skip_lo{}, std::true_type{}, no_x, buf + atap(1), buf + atap(2), NA, CA, wn + STEPB);
            kstep_a(no_skip{}, std::true_type{}, no_x, buf + atap(2), buf + atap(3), CA, NA, wn + 2 * STEPB);
            convert_store();
            kstep_a(no_skip{}, std::true_type{}, xload_low, buf + atap(3), buf + atap(4), NA, CA, wn + 3 * STEPB);
            kstep_a(no_skip{}, std::true_type{}, no_x, buf + atap(4), buf + atap(5), CA, NA, wn + 4 * STEPB);
            kstep_a(skip_hi{}, std::true_type{}, no_x, buf + atap(5), buf + atap(6), NA, CA, wn + 5 * STEPB);
            convert_low();
            // (no branch around MFMAs: hipcc copies the accumulators at every join.)  The sample's last k-step fetches the NEXT sample's first phase-B k-step, its
            // pz = 0 half, into NA (= T0: c0 / 8 is even); the pz = 1 half goes to T2 behind the last chunk
            kstep_a(skip_hi{}, std::false_type{}, no_x, buf + atap(6), buf + atap(6), CA, NA, more ? wn + 6 * STEPB : wB0);
            __syncthreads();
        };
        chunk_a(0, std::integral_constant<int, -2>{}, T0, T1);
        PP_STAMP(2);
        chunk_a(1, std::integral_constant<int, 1>{}, T1, T0);
        PP_STAMP(3);
        for (int ca = 2; ca < nA; ca += 2) {
            chunk_a(ca, std::integral_constant<int, -1>{}, T0, T1);
            chunk_a(ca + 1, std::integral_constant<int, -1>{}, T1, T0);
        }
        ldw(T2, wB1);
        PP_STAMP(5);

        // ---- epilogue, from registers: v = relu(hi + lo / 2^11); lane (g, v) holds couts 32 ch + 8 g + 4 j + r of voxel (vb, 2 rj + py, 2 ri + px)
        int te = tid;
        asm volatile("" : "+v"(te));
        const int el = te & 63, eg = el >> 4, nl = el & 15;
        if constexpr ((RF_PP_ABL & 2) != 0) {
            float sink = 0.f;
#pragma unroll
            for (int vb = 0; vb < 8; ++vb)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sink += hi[vb][j][r] + lo[vb][j][r];
            if (sink == 123.456f) a.pre_out[te][0] = (_Float16)sink;
            continue;
        }
#pragma unroll
        for (int vb = 0; vb < 8; ++vb)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) hi[vb][j][r] = fmaxf(fmaf(lo[vb][j][r], 1.0f / US_LO, hi[vb][j][r]), 0.f);
        double2* chst = reinterpret_cast<double2*>(lds + PP_CHST);
        double2* chs = reinterpret_cast<double2*>(lds + PP_CHS);
        float4* trip = reinterpret_cast<float4*>(lds + PP_TRIP);
        {
            // per cout: the lane's eight planes in float64, then over the 16 voxel lanes of the row: lanes n, n ^ 1 keep the total of value n >> 1 = 4 j + r
            double sm[8], sq[8];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double s_ = 0.0, q_ = 0.0;
#pragma unroll
                    for (int vb = 0; vb < 8; ++vb) {
                        const double v = (double)hi[vb][j][r];
                        s_ += v; q_ += v * v;
                    }
                    sm[j * 4 + r] = s_; sq[j * 4 + r] = q_;
                }
            const double ts = pp_row16_transpose_sum8(sm, nl), tq = pp_row16_transpose_sum8(sq, nl);
            if ((nl & 1) == 0) chst[(ch * 4 + q) * 32 + 8 * eg + (nl >> 1)] = make_double2(ts, tq);
        }
        PP_STAMP(6);
        __syncthreads();
        PP_STAMP(7);
        const int cout = a.cout;
        if (te < 256) {   // thread (channel te >> 2, column te & 3): the channel's sums over the four (py, px) waves (fixed order), lane 0 of the four publishes them
            const int c = te >> 2;
            double2 v = chst[((c >> 5) * 4 + (te & 3)) * 32 + (c & 31)];
            v.x += pp_dpp_f64<0xB1>(v.x); v.y += pp_dpp_f64<0xB1>(v.y);
            v.x += pp_dpp_f64<0x4E>(v.x); v.y += pp_dpp_f64<0x4E>(v.y);
            if ((te & 3) == 0) {
                chs[c] = v;
                if (a.stats && c < cout) a.stats[(size_t)n * cout + c] = v;
            }
        }
        __syncthreads();
        if (te < cout) {                                             // as rf_gn_from_stats: group sums in channel order, float64
            int cpg = cout / a.ngroups;
            float neps = a.neps;
            asm volatile("" : "+s"(cpg), "+s"(neps));              // (their float64 forms are not to live in VGPR pairs across the sample loop)
            const int cbeg = (te / cpg) * cpg;
            double sm = 0.0, sq = 0.0;
            for (int c = cbeg; c < cbeg + cpg; ++c) { sm += chs[c].x; sq += chs[c].y; }
            const double icnt = __hiloint2double(icnt_hi, icnt_lo), mean = sm * icnt;
            double var = sq * icnt - mean * mean;
            if (var < 0.0) var = 0.0;
            // 1 / sqrt in float64 by the hardware estimate + two Newton steps (the IEEE sqrt and division are ~100 dependent instructions between two barriers)
            const double xv = var + (double)neps;
            double rs = __builtin_amdgcn_rsq(xv);
            rs = rs * (1.5 - 0.5 * xv * rs * rs);
            rs = rs * (1.5 - 0.5 * xv * rs * rs);
            const float2 gb = reinterpret_cast<const float2*>(lds + PP_GB)[te];
            float4 t4 = gn_affine(mean, rs, gb.x, gb.y);
            t4.y *= US_ACT_SCALE; t4.z *= US_ACT_SCALE;              // the split's 2^-4, exact
            trip[te] = t4;
        }
        PP_STAMP(8);
        __syncthreads();
        PP_STAMP(9);
        {
            // slot of the lane's voxel (vb, 2 rj + py, 2 ri + px): linear vb 64 + y 8 + x, or parity-major ((vb & 1) 4 + 2 py + px) 64 + (vb >> 1) 16 + (lane & 15):
            // 256-byte runs per 8-channel group
            const int sg = 4 * ch + eg, nsg = cout >> 3;
            const int vox = a.pre_pm ? (2 * py + px) * 64 + nl : (2 * (nl >> 2) + py) * 8 + 2 * (nl & 3) + px;
            h8* __restrict__ po = a.pre_out + ((size_t)n * nsg + (sg < nsg ? sg : 0)) * 2 * 512 + vox;
            float4 t4[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t4[i] = trip[(sg < nsg ? sg : 0) * 8 + i];
#pragma unroll
            for (int vb = 0; vb < 8; ++vb) {
                float y[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) y[i] = fmaf(hi[vb][i >> 2][i & 3] - t4[i].x, t4[i].y, t4[i].z);
                h8 h, l;
                us_split8_scaled(y, h, l);
                const int so = a.pre_pm ? (vb & 1) * 256 + (vb >> 1) * 16 : vb * 64;
                if (sg < nsg) {
                    po[so] = h;
                    po[512 + so] = l;
                }
            }
        }
        PP_STAMP(10);
#ifdef RF_PP_STAMPS
        ++it;
#endif
    }
#ifdef RF_PP_STAMPS
    if (lane == 0 && (wave & 3) == 0 && blockIdx.x < 1024) {
#pragma unroll
        for (int i = 0; i < 16; ++i) g_pp_stamps[(blockIdx.x * 2 + (wave >> 2)) * 16 + i] = st[i];
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------- 4^3 volumes
// The decoder form on whole 4^3 samples (retrieval backbone dec0: 64 skip channels @4^3 + 128 channels upsampled from 2^3 -> 64): 8 samples
// per workgroup, wave = output parity (pz, py, px) as above, m-block m = one z plane (4 voxels) of the parity class of samples 4 (m >> 1) .. + 3
// (see the kernel), 16 couts per workgroup (grid.y).  Phase A (skip channels, 27 taps): 8 halo cubes of 6^3 slots, one voxel staged per thread and chunk.  Phase B
// (upsampled channels, the 8 pre-summed low-res taps of the weight image's B region): 8 halo cubes of 4^3 slots per chunk, four chunks
// staged at a time into the same LDS (the phase-A image is dead by then).  Epilogue through an LDS tile [cout][sample][64] as above.
namespace {
constexpr int U4_ASLOTS = 8 * 216, U4_A_PLANE = U4_ASLOTS * 16;           // 27,648
constexpr int U4_BSLOTS = 8 * 64, U4_B_PLANE = U4_BSLOTS * 16;            // 8,192 per chunk and piece
constexpr int U4_BG = 4;                                                  // phase-B chunks staged together
constexpr int U4_E_STRIDE = 516;
constexpr int U4_LDS_BYTES = 32 * U4_E_STRIDE * 4;                        // 66,048 (epilogue tile) >= 2 * U4_A_PLANE, >= U4_BG * 2 * U4_B_PLANE
static_assert(2 * U4_A_PLANE <= U4_LDS_BYTES && U4_BG * 2 * U4_B_PLANE <= U4_LDS_BYTES, "images must fit the epilogue tile's LDS");
}   // namespace

// PRE: weight fragments one k-step ahead in a second register set
template <int NB, bool PRE>
__global__ __launch_bounds__(512, NB >= 3 ? 2 : 4) void k_conv3_up_split_s4(UpSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = a.c0, c1 = a.c1, cin = c0 + c1, nA = c0 >> 3, nB = c1 >> 3;
    const int n0 = blockIdx.x * 8;
    const int nbt = (a.cout + 15) >> 4, nb0 = blockIdx.y * NB;
    const int pz = wave >> 2, py = (wave >> 1) & 1, px = wave & 1;
    const int g = lane >> 4, ri = lane & 15;

    auto zero_lds = [&](int bytes) {
        for (int i = tid; i < bytes / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0u, 0u, 0u, 0u);
    };

    f32x4 hi[4][NB], lo[4][NB];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { hi[m][nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; lo[m][nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // ---- m-blocks and skipped k-steps (round 4).  Row ri of m-block m: sample 4 (m >> 1) + (ri >> 2), parity-class voxel (qz, qy, qx) with (qy, qx) =
    // ((ri >> 1) & 1, ri & 1): an m-block lies in ONE z plane.  m & 1 = 0 is the parity class's BORDER plane (z = 0 for pz = 0, z = 3 for pz = 1: qz = pz),
    // m & 1 = 1 its interior plane (z = 2 / z = 1).  The border plane's taps that point out of the volume read padding: two of a chunk's seven k-steps
    // in phase A, one of the two in phase B -- skipped, 18 % of dec0's MFMAs.  WHICH k-steps depends on pz; two compiled variants behind a branch on pz
    // cost 70-140 spilled registers in this 256-VGPR kernel (hipcc hoists and sinks the variants' common code across the diamond), so there is one
    // variant and the waves with pz = 1 walk the taps in z-MIRRORED order (k-step s, lane group g: tap mirror(4 s + g), its weights gathered from the
    // fragment image per lane group; phase B: k-steps in the order tz = 1, 0): for every wave the padding k-steps of the border plane are the first ones.
    static_assert(PRE, "k_conv3_up_split_s4: the one-k-step-ahead weight prefetch is the only form kept");
    const int wstep = nbt * 128;
    const h8* const wA0 = a.wp + (size_t)nb0 * 128;
    const h8* const wB0 = wA0 + ((size_t)nA * 7 + (size_t)wave * nB * 2) * wstep;
    auto mirror = [](int i) { return i < 27 ? (2 - i / 9) * 9 + i % 9 : 27; };
    auto sel4 = [&](int v0, int v1, int v2, int v3) { const int a01 = g & 1 ? v1 : v0, a23 = g & 1 ? v3 : v2; return g & 2 ? a23 : a01; };
    // this lane's weight fragment of phase-A k-step (ca, s): [h | l at + 64]
    auto pA = [&](int ca, int s) -> const h8* {
        const h8* cb = wA0 + (size_t)(ca * 7) * wstep;
        const int t0 = mirror(4 * s), t1 = mirror(4 * s + 1), t2 = mirror(4 * s + 2), t3 = mirror(4 * s + 3);
        const int offm = sel4((t0 >> 2) * wstep + (t0 & 3) * 16, (t1 >> 2) * wstep + (t1 & 3) * 16, (t2 >> 2) * wstep + (t2 & 3) * 16, (t3 >> 2) * wstep + (t3 & 3) * 16) + ri;
        return cb + (pz ? offm : s * wstep + lane);
    };
    auto pB = [&](int cgi, int tzp) -> const h8* { return wB0 + (size_t)(cgi * 2 + (tzp ^ pz)) * wstep + lane; };
    h8 bh[NB], bl[NB], nh[NB], nl[NB];
    auto load_p = [&](const h8* p, h8 (&h)[NB], h8 (&l)[NB]) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { h[nb] = p[nb * 128]; l[nb] = p[nb * 128 + 64]; }
    };
    load_p(nA ? pA(0, 0) : pB(0, 0), bh, bl);

    // ---------------------------------------------------------------- phase A
    {
        const int sw = tid >> 6;                                    // staging: this thread's voxel `lane` of sample sw
        const int ns = n0 + sw < a.n ? n0 + sw : a.n - 1;           // ragged last group: re-reads the last sample, its stores are masked
        const float4* __restrict__ aff = a.affine + (size_t)ns * cin;
        const float* __restrict__ s0 = a.src0 + (size_t)ns * c0 * 64 + lane;
        unsigned char* const myslot = lds + (sw * 216 + ((lane >> 4) + 1) * 36 + (((lane >> 2) & 3) + 1) * 6 + (lane & 3) + 1) * 16;
        auto stage_load = [&](float (&x)[8], int ca) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = s0[(size_t)(ca * 8 + j) * 64];
        };
        auto stage_store = [&](const float (&x)[8], int ca) {
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 af = aff[ca * 8 + j];
                y[j] = fmaf(x[j] - af.x, af.y, af.z);
            }
            h8 h, l;
            us_split8(y, h, l);
            *reinterpret_cast<h8*>(myslot) = h;
            *reinterpret_cast<h8*>(myslot + U4_A_PLANE) = l;
        };
        // (z, y, x) = (2 qz + pz, 2 qy + py, 2 qx + px) in the sample's 6^3 halo cube (+1): the border plane is z = 3 pz, the interior plane z = 2 - pz
        const unsigned char* const abase = lds + ((ri >> 2) * 216 + (2 * ((ri >> 1) & 1) + py + 1) * 6 + 2 * (ri & 1) + px + 1) * 16;
        const int zplane[2] = {(3 * pz + 1) * 36 * 16, (3 - pz) * 36 * 16};
        auto tapoff = [](int tp) { tp = tp < 27 ? tp : 26; return ((tp / 9 - 1) * 36 + ((tp / 3) % 3 - 1) * 6 + (tp % 3 - 1)) * 16; };       // tap 27: zero weights, reads tap 26
        float xr[8];
        if (nA) stage_load(xr, 0);
        zero_lds(2 * U4_A_PLANE);                                   // the rings are the zero padding and are written once
        __syncthreads();
        if (nA) stage_store(xr, 0);
        __syncthreads();
        for (int ca = 0; ca < nA; ++ca) {
            const bool more = ca + 1 < nA;
            if (more) stage_load(xr, ca + 1);
#pragma unroll
            for (int s = 0; s < 7; ++s) {
                load_p(s < 6 ? pA(ca, s + 1) : (more ? pA(ca + 1, 0) : pB(0, 0)), nh, nl);
                // the lane group's tap offset of this k-step: compile-time constants selected by g and pz
                const int atn = sel4(tapoff(4 * s), tapoff(4 * s + 1), tapoff(4 * s + 2), tapoff(4 * s + 3));
                const int atm = sel4(tapoff(mirror(4 * s)), tapoff(mirror(4 * s + 1)), tapoff(mirror(4 * s + 2)), tapoff(mirror(4 * s + 3)));
                const int at = pz ? atm : atn;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if ((m & 1) == 0 && s < 2) continue;            // border plane, taps 0..7 of this wave's order: all out of the volume
                    const unsigned char* ap = abase + (m >> 1) * (4 * 216 * 16) + zplane[m & 1] + at;
                    const h8 ah = *reinterpret_cast<const h8*>(ap);
                    const h8 al = *reinterpret_cast<const h8*>(ap + U4_A_PLANE);
                    us_mfma_block<NB>(hi[m], lo[m], ah, al, bh, bl);
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) { bh[nb] = nh[nb]; bl[nb] = nl[nb]; }
            }
            __syncthreads();
            if (more) {
                stage_store(xr, ca + 1);
                __syncthreads();
            }
        }
    }

    // ---------------------------------------------------------------- phase B: [chunk in group][sample][4^3 halo cube], h and l plane per chunk
    {
        // staging item (thread < 64 * U4_BG): chunk-in-group tid >> 6, sample (tid >> 3) & 7, low-res voxel tid & 7
        const int scg = tid >> 6, ssm = (tid >> 3) & 7, sv = tid & 7;
        const int ns = n0 + ssm < a.n ? n0 + ssm : a.n - 1;
        const float4* __restrict__ aff = a.affine + (size_t)ns * cin + c0;
        const float* __restrict__ s1 = a.src1 + (size_t)ns * c1 * 8 + sv;
        unsigned char* const myslot = lds + scg * 2 * U4_B_PLANE + (ssm * 64 + ((sv >> 2) + 1) * 16 + (((sv >> 1) & 1) + 1) * 4 + (sv & 1) + 1) * 16;
        // k-step tz, lane group (ty, tx): low-res halo voxel (qz + tz + pz, qy + ty + py, qx + tx + px); halo planes 0 and 3 are padding.  Border plane
        // (qz = pz): halo z = 2 pz + tz, padding for tz = pz -- this wave's first k-step (tz = tzp ^ pz); interior plane (qz = 1 - pz): halo z = 1 + tz
        const unsigned char* const bbase = lds + ((ri >> 2) * 64 + (((ri >> 1) & 1) + py + (g >> 1)) * 4 + (ri & 1) + px + (g & 1)) * 16;
        const int zb[2][2] = {{(2 * pz + pz) * 256, (1 + pz) * 256}, {(2 * pz + (1 ^ pz)) * 256, (1 + (1 ^ pz)) * 256}};      // [tzp][m & 1]
        zero_lds(U4_BG * 2 * U4_B_PLANE);                           // (phase A ended on a barrier)
        __syncthreads();
        for (int cb0 = 0; cb0 < nB; cb0 += U4_BG) {
            const int ng = nB - cb0 < U4_BG ? nB - cb0 : U4_BG;
            if (cb0) __syncthreads();                               // everyone left the previous group
            if (scg < ng) {
                float y[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const float4 af = aff[(cb0 + scg) * 8 + jj];
                    y[jj] = fmaf(s1[(size_t)((cb0 + scg) * 8 + jj) * 8] - af.x, af.y, af.z);
                }
                h8 h, l;
                us_split8(y, h, l);
                *reinterpret_cast<h8*>(myslot) = h;
                *reinterpret_cast<h8*>(myslot + U4_B_PLANE) = l;
            }
            __syncthreads();
            for (int cg = 0; cg < ng; ++cg) {
                const int cgi = cb0 + cg;
#pragma unroll
                for (int tzp = 0; tzp < 2; ++tzp) {
                    load_p(tzp == 0 ? pB(cgi, 1) : pB(cgi + 1 < nB ? cgi + 1 : cgi, 0), nh, nl);       // (after the last k-step: any fragment of the image)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        if ((m & 1) == 0 && tzp == 0) continue;     // border plane, the k-step whose low-res plane is outside
                        const unsigned char* p = bbase + cg * 2 * U4_B_PLANE + (m >> 1) * (4 * 64 * 16) + zb[tzp][m & 1];
                        const h8 ah = *reinterpret_cast<const h8*>(p);
                        const h8 al = *reinterpret_cast<const h8*>(p + U4_B_PLANE);
                        us_mfma_block<NB>(hi[m], lo[m], ah, al, bh, bl);
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) { bh[nb] = nh[nb]; bl[nb] = nl[nb]; }
                }
            }
        }
    }
    __syncthreads();

    // ---------------------------------------------------------------- epilogue: relu(hi + lo / 2^11) -> LDS tile [cout][sample][64] -> float4 rows
    float* e = reinterpret_cast<float*>(lds);
    {
        const int col = lane & 15;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * g + r, sm = 4 * (m >> 1) + (row >> 2);
                    const int z = m & 1 ? 2 - pz : 3 * pz;
                    const int lin = sm * 64 + z * 16 + (2 * ((row >> 1) & 1) + py) * 4 + 2 * (row & 1) + px;
                    e[(nb * 16 + col) * U4_E_STRIDE + lin] = fmaxf(fmaf(lo[m][nb][r], 1.0f / US_LO, hi[m][nb][r]), 0.f);
                }
    }
    __syncthreads();
    const int cout = a.cout, cob = nb0 * 16;
    for (int qd = tid; qd < NB * 16 * 128; qd += 512) {             // (cout, sample, float4): 16 float4 per (cout, sample)
        const int co = qd >> 7, sm = (qd >> 4) & 7, l4 = qd & 15;
        if (cob + co < cout && n0 + sm < a.n)
            *reinterpret_cast<float4*>(a.out + ((size_t)(n0 + sm) * cout + cob + co) * 64 + l4 * 4) =
                *reinterpret_cast<const float4*>(e + co * U4_E_STRIDE + sm * 64 + l4 * 4);
    }
    if (a.stats) {
        // per (cout, sample): two threads sum 32 values each (float64), then the two partial sums; 32 couts per pass of the 512 threads
#pragma unroll
        for (int cpass = 0; cpass < NB * 16; cpass += 32) {
            const int co = cpass + (tid >> 4), sm = (tid >> 1) & 7, part = tid & 1;
            double s = 0.0, sq = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (co >= NB * 16) break;
                const float4 v = *reinterpret_cast<const float4*>(e + co * U4_E_STRIDE + sm * 64 + (part * 8 + i) * 4);
                s += (double)v.x; sq += (double)v.x * v.x;
                s += (double)v.y; sq += (double)v.y * v.y;
                s += (double)v.z; sq += (double)v.z * v.z;
                s += (double)v.w; sq += (double)v.w * v.w;
            }
            s += __shfl_xor(s, 1, 64); sq += __shfl_xor(sq, 1, 64);
            if (part == 0 && co < NB * 16 && cob + co < cout && n0 + sm < a.n) a.stats[(size_t)(n0 + sm) * cout + cob + co] = make_double2(s, sq);
        }
    }
}

// -------------------------------------------------------------------------------------------------------------------- host
// ------------------------------------------------------------------------------------- box tiles of a large volume, no skip source
// DecoderNoJoining's first conv (model/unet.py:311-322: x2 nearest upsample, GroupNorm -> conv3 -> ReLU; the final decoder's 16 -> 16 @64^3 from
// 32^3, the U-Net backbone's 32 -> 32 @16^3 / 32 -> 16 @32^3): only upsampled channels, so only phase B of the kernel above -- the 8 pre-summed
// low-resolution taps per output parity.  A workgroup owns one 8^3 output box of an edge^3 volume: its low-res halo box is 6^3 voxels of the
// half-resolution source (zeros outside the volume), all channel groups staged at once (thread = (group, halo voxel)); wave = output parity,
// m-block m = z pair m; per channel group two k-steps (tz = 0 / 1), lane group = (ty, tx).  With 16 input channels that is 48 MFMAs per wave:
// the kernel is its staging (216 x c1 values in) and its epilogue (512 x cout values out + statistics), and small enough in registers and
// LDS (33 KB) for four workgroups per CU to overlap them.
namespace {
constexpr int UB_E_STRIDE = 516;
}
template <int NB>
__global__ __launch_bounds__(512, 4) void k_conv3_up_split_box(UpSplitArgs a, int edge) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pz = wave >> 2, py = (wave >> 1) & 1, px = wave & 1;
    const int c1 = a.c1, nB = c1 >> 3, half = edge >> 1, tpe = edge >> 3;
    // XCD-contiguous box order (conv_box.h): neighbouring boxes share halo voxels and output cache lines
    const unsigned g_ = gridDim.x, per = g_ >> 3, rem = g_ & 7u, kx = blockIdx.x & 7u;
    int t = (int)(kx * per + (kx < rem ? kx : rem) + (blockIdx.x >> 3));
    const int tile = t % (tpe * tpe * tpe);
    const int x0 = (t % tpe) * 8; t /= tpe;
    const int y0 = (t % tpe) * 8; t /= tpe;
    const int z0 = (t % tpe) * 8; t /= tpe;
    const int n = t;
    const float4* __restrict__ aff = a.affine + (size_t)n * c1;
    const size_t hvol = (size_t)half * half * half;

    // ---- stage: thread = (channel group, halo voxel); 216 voxels per group
    for (int u = tid; u < nB * US_BSLOTS; u += 512) {
        const int cg = u / US_BSLOTS, v = u % US_BSLOTS;
        const int hz = v / US_BZ, hy = (v / US_BY) % 6, hx = v % 6;
        const int z = (z0 >> 1) + hz - 1, y = (y0 >> 1) + hy - 1, x = (x0 >> 1) + hx - 1;
        const bool in = (unsigned)z < (unsigned)half && (unsigned)y < (unsigned)half && (unsigned)x < (unsigned)half;
        float yv[8];
        const float* __restrict__ sp = a.src1 + ((size_t)n * c1 + cg * 8) * hvol + (in ? ((size_t)z * half + y) * half + x : 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 af = aff[cg * 8 + j];
            yv[j] = in ? fmaf(sp[(size_t)j * hvol] - af.x, af.y, af.z) : 0.f;       // zero padding of the NORMALISED tensor
        }
        h8 h, l;
        us_split8(yv, h, l);
        unsigned char* p = lds + cg * 2 * US_B_PLANE + v * 16;
        *reinterpret_cast<h8*>(p) = h;
        *reinterpret_cast<h8*>(p + US_B_PLANE) = l;
    }

    const int g = lane >> 4, rj = (lane >> 2) & 3, ri = lane & 3;
    const int bbase = (pz * US_BZ + (rj + py + (g >> 1)) * US_BY + (ri + px + (g & 1))) * 16;        // + (m + tz) BZ
    f32x4 hi[4][NB], lo[4][NB];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { hi[m][nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; lo[m][nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    constexpr int STEP_U4 = NB * 2 * 64;
    const h8* __restrict__ wn = a.wp + (size_t)wave * nB * 2 * STEP_U4 + lane;       // this parity's stream: [group][tz][nb][h|l][64]
    h8 bh[NB], bl[NB];
    __syncthreads();
    for (int cb = 0; cb < nB; ++cb) {
#pragma unroll
        for (int tz = 0; tz < 2; ++tz) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) { bh[nb] = wn[(nb * 2) * 64]; bl[nb] = wn[(nb * 2 + 1) * 64]; }
            wn += STEP_U4;
            const unsigned char* ap = lds + cb * 2 * US_B_PLANE + bbase + tz * US_BZ * 16;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const h8 ah = *reinterpret_cast<const h8*>(ap + m * US_BZ * 16);
                const h8 al = *reinterpret_cast<const h8*>(ap + m * US_BZ * 16 + US_B_PLANE);
                us_mfma_block<NB>(hi[m], lo[m], ah, al, bh, bl);
            }
        }
    }
    __syncthreads();

    // ---- epilogue: relu(hi + lo / 2^11) -> LDS tile [cout][z][y][x] of the box -> float4 half rows, statistics of the box per cout
    float* e = reinterpret_cast<float*>(lds);
    {
        const int col = lane & 15, yj = lane >> 4;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lin = (2 * m + pz) * 64 + (2 * yj + py) * 8 + 2 * r + px;
                    e[(nb * 16 + col) * UB_E_STRIDE + lin] = fmaxf(fmaf(lo[m][nb][r], 1.0f / US_LO, hi[m][nb][r]), 0.f);
                }
    }
    __syncthreads();
    const int cout = a.cout;
    const size_t vol = (size_t)edge * edge * edge;
    if (a.out_ch8) {
        // channel-interleaved output for a consumer that stages 8 channels of a voxel at a time: a box row is 8 voxels x 32 bytes = 256 contiguous bytes
        float* __restrict__ oc = a.out + (size_t)n * cout * vol;
        for (int q = tid; q < (cout >> 3) * 1024; q += 512) {
            const int cg = q >> 10, vox = (q >> 1) & 511, hf = q & 1;
            const int z = vox >> 6, y = (vox >> 3) & 7, x = vox & 7;
            const float* ep = e + (cg * 8 + hf * 4) * UB_E_STRIDE + vox;
            *reinterpret_cast<float4*>(oc + (((size_t)cg * vol + ((size_t)(z0 + z) * edge + y0 + y) * edge + x0 + x) << 3) + hf * 4) =
                make_float4(ep[0], ep[UB_E_STRIDE], ep[2 * UB_E_STRIDE], ep[3 * UB_E_STRIDE]);
        }
    }
    float* __restrict__ o = a.out + (size_t)n * cout * vol + ((size_t)z0 * edge + y0) * edge + x0;
    for (int q = tid; q < (a.out_ch8 ? 0 : cout * 128); q += 512) {
        const int co = q >> 7, l4 = q & 127;                          // l4: float4 index inside the box: (z, y, half row)
        const int z = l4 >> 4, y = (l4 >> 1) & 7, xh = l4 & 1;
        *reinterpret_cast<float4*>(o + (size_t)co * vol + ((size_t)z * edge + y) * edge + xh * 4) = *reinterpret_cast<const float4*>(e + co * UB_E_STRIDE + l4 * 4);
    }
    if (a.stats) {
        const int tiles = tpe * tpe * tpe;
        const int co = tid >> 3, part = tid & 7;
        double sm = 0.0, sq = 0.0;
        if (co < cout) {
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(e + co * UB_E_STRIDE + (part * 16 + i) * 4);
                sm += (double)v.x; sq += (double)v.x * v.x;
                sm += (double)v.y; sq += (double)v.y * v.y;
                sm += (double)v.z; sq += (double)v.z * v.z;
                sm += (double)v.w; sq += (double)v.w * v.w;
            }
        }
#pragma unroll
        for (int msk = 1; msk < 8; msk <<= 1) { sm += __shfl_xor(sm, msk, 64); sq += __shfl_xor(sq, msk, 64); }
        if (part == 0 && co < cout) a.stats[((size_t)n * cout + co) * tiles + tile] = make_double2(sm, sq);
    }
}

// ------------------------------------------------------------------------------ the box kernel, persistent
// k_conv3_up_split_box for the widest volumes (the final decoder's 16 -> 16 @64^3: 16,384 boxes per step of 32 chunks): the layer has 48 MFMAs per
// wave and box, so a workgroup per box spends its life staging, storing and waiting at its own start and end.  Here a workgroup (512 threads, two per
// CU) owns a run of boxes (every 64th of its XCD's range, see the kernel): each wave keeps its parity's weights in registers for the whole run, the low-res halo image is double-buffered
// (box i + 2 is requested while box i's outputs are stored, box i + 1 converted while the other waves finish box i's MFMAs), and the epilogue tile is
// drained while the next voxels are in flight.  Two barriers per box:
//     MFMA(i) | A | tile(i) <- accumulators, image(i + 1) <- registers | B | request box i + 2, store tile(i), statistics(i)
// Results are those of k_conv3_up_split_box bit for bit (same products, same accumulation order, same statistics order).
// Measured (tools/upbox_bench.py, tools/upbox_ablation.py; 32 chunks): 0.315 -> 0.25 ms alone.  Without the voxel loads 0.17, without the stores 0.17,
// with neither 0.09: a box's loads and stores go out in two bursts and only two workgroups per CU interleave them.  A rotated loop whose waits for
// the voxels leave the stores in flight (vmcnt(4) instead of vmcnt(0)) and a two-boxes-deep request measured 0.259 / 0.268: not kept.
namespace {
constexpr int UP_IMG = 2 * 2 * US_B_PLANE;                         // one halo image: <= 2 channel groups x (h | l)            13,824
constexpr int UP_TILE = 2 * UP_IMG;                                // epilogue tile [16][UB_E_STRIDE] fp32                      33,024
constexpr int UP_AFF = UP_TILE + 16 * UB_E_STRIDE * 4;             // <= 16 GroupNorm triples of the sample being staged
constexpr int UP_LDS_BYTES = UP_AFF + 16 * 16;                     // 60,928: two workgroups per CU
}
template <int NBG, int CGO>
__global__ __launch_bounds__(512, 4) void k_conv3_up_split_boxp(UpSplitArgs a, int edge, int total_boxes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pz = wave >> 2, py = (wave >> 1) & 1, px = wave & 1;
    constexpr int c1 = NBG * 8;
    const int half = edge >> 1, tpe = edge >> 3;
    const int lt = 31 - __builtin_clz(tpe);                        // edge is a power of two
    const int hvol = half * half * half;
    constexpr int cout = CGO * 8;                                  // channel-interleaved output only: whole groups of 8
    const int vol = edge * edge * edge;
    // Which boxes: XCD k (workgroups k, k + 8, ...: the dispatcher deals workgroups round-robin over the 8 XCDs) owns the k-th eighth of the boxes, and its
    // workgroups walk that range TOGETHER -- workgroup j of the XCD takes boxes j, j + bs, j + 2 bs, ... of it (bs = workgroups per XCD).  At any time an
    // XCD therefore works on ~bs consecutive boxes (a z slab of a sample): the low-res rows (128-byte lines that 8 x-neighbours and the y / z halos
    // share) are fetched into its L2 once.  With a contiguous run per workgroup they were fetched again for nearly every box (the workgroup's own
    // output stream had evicted them): 849 MB read per launch for a 67 MB source (PMC, profiles/r04_pmc_bench_C2_B32.csv).
    const int bs = (int)(gridDim.x >> 3), per_xcd = (total_boxes + 7) >> 3;
    const int b0 = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3), b1 = min(((int)(blockIdx.x & 7u) + 1) * per_xcd, total_boxes);
    if (b0 >= b1) return;

    // this thread's staging item: (channel group, halo voxel) -- one item per thread, NBG x 216 of the 512 threads have one
    const bool stager = tid < NBG * US_BSLOTS;
    const int scg = tid / US_BSLOTS, sv = tid % US_BSLOTS;
    const int shz = sv / US_BZ - 1, shy = (sv / US_BY) % 6 - 1, shx = sv % 6 - 1;
    float xr[8];
    bool xin = false;
    // raw voxels of box b -> registers.  The GroupNorm triples sit in an LDS table that is rewritten only when a run crosses into the next sample
    // (refresh): a per-box load of them would put a vmcnt(0) -- the round trip of these loads AND of the previous box's stores -- into every box
    auto request = [&](int b) {
        const int n = b >> (3 * lt), t = b & ((1 << (3 * lt)) - 1);
        const int z = ((t >> (2 * lt)) << 2) + shz, y = (((t >> lt) & (tpe - 1)) << 2) + shy, x = ((t & (tpe - 1)) << 2) + shx;
        xin = stager && (unsigned)z < (unsigned)half && (unsigned)y < (unsigned)half && (unsigned)x < (unsigned)half;
        const float* __restrict__ sb = a.src1 + (size_t)n * c1 * hvol;                  // uniform base, 32-bit lane offsets
        const unsigned off = xin ? (unsigned)(scg * 8 * hvol + (z * half + y) * half + x) : 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[j] = sb[off + (unsigned)(j * hvol)];
    };
    auto refresh = [&](int n) {
        if (tid < c1) *reinterpret_cast<float4*>(lds + UP_AFF + tid * 16) = a.affine[(size_t)n * c1 + tid];
    };
    auto stage = [&](int buf) {                                     // registers -> normalised, split halo image
        if (!stager) return;
        float yv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 af = *reinterpret_cast<const float4*>(lds + UP_AFF + (scg * 8 + j) * 16);
            yv[j] = xin ? fmaf(xr[j] - af.x, af.y, af.z) : 0.f;                          // zero padding of the NORMALISED tensor
        }
        h8 h, l;
        us_split8(yv, h, l);
        unsigned char* p = lds + buf * UP_IMG + scg * 2 * US_B_PLANE + sv * 16;
        *reinterpret_cast<h8*>(p) = h;
        *reinterpret_cast<h8*>(p + US_B_PLANE) = l;
    };

    // ---- this parity's weights: [group][tz][h | l], 16 VGPRs per group, resident for the whole run
    h8 wh[NBG][2], wl[NBG][2];
    {
        const h8* __restrict__ wn = a.wp + (size_t)wave * NBG * 2 * 128 + lane;
#pragma unroll
        for (int cb = 0; cb < NBG; ++cb)
#pragma unroll
            for (int tz = 0; tz < 2; ++tz) { wh[cb][tz] = wn[(cb * 2 + tz) * 128]; wl[cb][tz] = wn[(cb * 2 + tz) * 128 + 64]; }
    }
    const int g = lane >> 4, rj = (lane >> 2) & 3, ri = lane & 3;
    const int bbase = (pz * US_BZ + (rj + py + (g >> 1)) * US_BY + (ri + px + (g & 1))) * 16;        // + (m + tz) BZ
    float* const e = reinterpret_cast<float*>(lds + UP_TILE);

    request(b0);
    refresh(b0 >> (3 * lt));
    __syncthreads();
    stage(0);
    if (b0 + bs < b1) request(b0 + bs);
    __syncthreads();

    int cur = 1;
    for (int b = b0; b < b1; b += bs) {
        cur ^= 1;
        f32x4 hi[4][1], lo[4][1];
#pragma unroll
        for (int m = 0; m < 4; ++m) { hi[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; lo[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int cb = 0; cb < NBG; ++cb)
#pragma unroll
            for (int tz = 0; tz < 2; ++tz) {
                const unsigned char* ap = lds + cur * UP_IMG + cb * 2 * US_B_PLANE + bbase + tz * US_BZ * 16;
                const h8 (&bh)[1] = reinterpret_cast<const h8 (&)[1]>(wh[cb][tz]);
                const h8 (&bl)[1] = reinterpret_cast<const h8 (&)[1]>(wl[cb][tz]);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const h8 ah = *reinterpret_cast<const h8*>(ap + m * US_BZ * 16);
                    const h8 al = *reinterpret_cast<const h8*>(ap + m * US_BZ * 16 + US_B_PLANE);
                    us_mfma_block<1>(hi[m], lo[m], ah, al, bh, bl);
                }
            }
        if (b + bs < b1 && ((b + bs) >> (3 * lt)) != (b >> (3 * lt))) refresh((b + bs) >> (3 * lt));   // the next box is in another sample (the table was last read before B(b - 1))
        __syncthreads();                                           // A: tile(b - 1) drained by every wave, image(cur ^ 1) free since MFMA(b - 1)
        {
            const int col = lane & 15, yj = lane >> 4;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lin = (2 * m + pz) * 64 + (2 * yj + py) * 8 + 2 * r + px;
                    e[col * UB_E_STRIDE + lin] = fmaxf(fmaf(lo[m][0][r], 1.0f / US_LO, hi[m][0][r]), 0.f);
                }
        }
        if (b + bs < b1) stage(cur ^ 1);
        __syncthreads();                                           // B: tile(b) and image(b + 1) complete
        if (b + 2 * bs < b1) request(b + 2 * bs);

        const int n = b >> (3 * lt), tile = b & ((1 << (3 * lt)) - 1);
        const int z0 = (tile >> (2 * lt)) << 3, y0 = ((tile >> lt) & (tpe - 1)) << 3, x0 = (tile & (tpe - 1)) << 3;
        float* __restrict__ on = a.out + (size_t)n * cout * vol;                          // uniform; lane offsets are 32-bit (a sample is < 2^28 floats)
        // channel-interleaved: a box row is 8 voxels x 32 bytes = 256 contiguous bytes.  A compile-time number of stores: the waits for the NEXT box's
        // voxels (requested above, retired in order before these) can then leave the stores in flight
#pragma unroll
        for (int it = 0; it < 2 * CGO; ++it) {
            const int q = tid + it * 512;
            const int cg = q >> 10, vox = (q >> 1) & 511, hf = q & 1;
            const int z = vox >> 6, y = (vox >> 3) & 7, x = vox & 7;
            const float* ep = e + (cg * 8 + hf * 4) * UB_E_STRIDE + vox;
            const unsigned o = ((unsigned)(cg * vol + ((z0 + z) * edge + y0 + y) * edge + x0 + x) << 3) + hf * 4;
            *reinterpret_cast<float4*>(on + o) = make_float4(ep[0], ep[UB_E_STRIDE], ep[2 * UB_E_STRIDE], ep[3 * UB_E_STRIDE]);
        }
        if (a.stats) {
            const int co = tid >> 3, part = tid & 7;
            double sm = 0.0, sq = 0.0;
            if (co < cout) {
#pragma unroll 4
                for (int i = 0; i < 16; ++i) {
                    const float4 v = *reinterpret_cast<const float4*>(e + co * UB_E_STRIDE + (part * 16 + i) * 4);
                    sm += (double)v.x; sq += (double)v.x * v.x;
                    sm += (double)v.y; sq += (double)v.y * v.y;
                    sm += (double)v.z; sq += (double)v.z * v.z;
                    sm += (double)v.w; sq += (double)v.w * v.w;
                }
            }
#pragma unroll
            for (int msk = 1; msk < 8; msk <<= 1) { sm += __shfl_xor(sm, msk, 64); sq += __shfl_xor(sq, msk, 64); }
            if (part == 0 && co < cout) a.stats[((size_t)n * cout + co) * (size_t)(1 << (3 * lt)) + tile] = make_double2(sm, sq);
        }
    }
}

// ------------------------------------------------------------------------------ 8^3 boxes of a large volume WITH a skip source
// The decoder form for the U-Nets' decoder stages on 16^3 ... 128^3 volumes (C5's 48 + 96 -> 78 @32^3, reference model/refinement.py:37-45 +
// model/unet.py:297-308): phase A as k_conv3_up_split (27 taps on the full-res halo box, 7 k-steps per 8-channel chunk) with the halo read from the
// neighbouring boxes (zero outside the volume: the padding of the NORMALISED tensor), phase B as k_conv3_up_split_box.  One workgroup per (box, group
// of NB n-blocks): wave = output parity, 4 m-blocks x NB n-blocks.  LDS: one phase-A chunk image (38.6 KB) + all low-res groups (6.9 KB each), the
// epilogue tile aliases both.  A chunk is staged by all threads (two halo voxels each), then its seven k-steps run: no double buffer -- the
// layers this kernel serves have a few thousand boxes, not the retrieval backbone's 8192 x 32.
namespace {
constexpr int UK_A_OFF = 0, UK_B_OFF = US_A_BUF;                   // chunk image, then the low-res groups
}
template <int NB>
__global__ __launch_bounds__(512, 2) void k_conv3_up_split_boxskip(UpSplitArgs a, int edge) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pz = wave >> 2, py = (wave >> 1) & 1, px = wave & 1;
    const int c0 = a.c0, c1 = a.c1, cin = c0 + c1, nA = c0 >> 3, nB = c1 >> 3, half = edge >> 1, tpe = edge >> 3;
    const int nbt = (a.cout + 15) >> 4;                            // n-blocks of the weight image
    const int nb0 = blockIdx.y * NB;                                // this workgroup's first n-block
    const unsigned g_ = gridDim.x, per = g_ >> 3, rem = g_ & 7u, kx = blockIdx.x & 7u;
    int t = (int)(kx * per + (kx < rem ? kx : rem) + (blockIdx.x >> 3));
    const int tile = t % (tpe * tpe * tpe);
    const int x0 = (t % tpe) * 8; t /= tpe;
    const int y0 = (t % tpe) * 8; t /= tpe;
    const int z0 = (t % tpe) * 8; t /= tpe;
    const int n = t;
    const float4* __restrict__ aff = a.affine + (size_t)n * cin;
    const size_t vol = (size_t)edge * edge * edge, hvol = (size_t)half * half * half;

    // ---- zero the chunk image's stride padding once (slots the staging never writes are never read either, but keep them defined)
    for (int i = tid; i < US_A_BUF / 16; i += 512) reinterpret_cast<uint4*>(lds + UK_A_OFF)[i] = make_uint4(0u, 0u, 0u, 0u);

    // ---- low-res groups: thread = (channel group, halo voxel); 216 voxels per group
    for (int u = tid; u < nB * US_BSLOTS; u += 512) {
        const int cg = u / US_BSLOTS, v = u % US_BSLOTS;
        const int hz = v / US_BZ, hy = (v / US_BY) % 6, hx = v % 6;
        const int z = (z0 >> 1) + hz - 1, y = (y0 >> 1) + hy - 1, x = (x0 >> 1) + hx - 1;
        const bool in = (unsigned)z < (unsigned)half && (unsigned)y < (unsigned)half && (unsigned)x < (unsigned)half;
        float yv[8];
        const float* __restrict__ sp = a.src1 + ((size_t)n * c1 + cg * 8) * hvol + (in ? ((size_t)z * half + y) * half + x : 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 af = aff[c0 + cg * 8 + j];
            yv[j] = in ? fmaf(sp[(size_t)j * hvol] - af.x, af.y, af.z) : 0.f;
        }
        h8 h, l;
        us_split8(yv, h, l);
        unsigned char* p = lds + UK_B_OFF + cg * 2 * US_B_PLANE + v * 16;
        *reinterpret_cast<h8*>(p) = h;
        *reinterpret_cast<h8*>(p + US_B_PLANE) = l;
    }

    // ---- phase-A staging: thread owns halo voxels tid and tid + 512 (the second only below 1000) of the 10^3 box
    int voff[2], vsl[2];
    bool vin[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int v = tid + r * 512;
        const int hx = v % 10, hy = (v / 10) % 10, hz = v / 100;
        const int z = z0 + hz - 1, y = y0 + hy - 1, x = x0 + hx - 1;
        vin[r] = v < 1000 && (unsigned)z < (unsigned)edge && (unsigned)y < (unsigned)edge && (unsigned)x < (unsigned)edge;
        voff[r] = vin[r] ? (z * edge + y) * edge + x : 0;
        vsl[r] = v < 1000 ? hz * US_SZ + hy * US_SY + hx : -1;
    }
    const float* __restrict__ sb0 = a.src0 + (size_t)n * c0 * vol;

    const int g = lane >> 4, rj = (lane >> 2) & 3, ri = lane & 3;
    const int abase = ((pz + 1) * US_SZ + (2 * rj + py + 1) * US_SY + (2 * ri + px + 1)) * 16;
    int atap[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int tp = 4 * s + g < 27 ? 4 * s + g : 26;
        atap[s] = ((tp / 9 - 1) * US_SZ + ((tp / 3) % 3 - 1) * US_SY + (tp % 3 - 1)) * 16;
    }
    const int bbase = (pz * US_BZ + (rj + py + (g >> 1)) * US_BY + (ri + px + (g & 1))) * 16;

    f32x4 hi[4][NB], lo[4][NB];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { hi[m][nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; lo[m][nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // weight image: [k-step][n-block of nbt][h | l][64 lanes]; n-blocks past the image's last one re-read it (their couts are masked at the store)
    const int wstep = nbt * 128;
    int nbo[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) nbo[nb] = (nb0 + nb < nbt ? nb0 + nb : nbt - 1) * 128;
    const h8* __restrict__ wA = a.wp + lane;
    const h8* __restrict__ wB = a.wp + (size_t)nA * 7 * wstep + (size_t)wave * nB * 2 * wstep + lane;
    h8 bh[NB], bl[NB], nh[NB], nl[NB];
    auto load_w = [&](const h8* w, h8 (&h)[NB], h8 (&l)[NB]) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { h[nb] = w[nbo[nb]]; l[nb] = w[nbo[nb] + 64]; }
    };
    load_w(nA > 0 ? wA : wB, bh, bl);

    // ---- phase A
    for (int ca = 0; ca < nA; ++ca) {
        float x[2][8];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[r][j] = sb0[(size_t)(ca * 8 + j) * vol + voff[r]];
        __syncthreads();                                           // the previous chunk's k-steps are done with the image
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 af = aff[ca * 8 + j];
                y[j] = vin[r] ? fmaf(x[r][j] - af.x, af.y, af.z) : 0.f;
            }
            h8 h, l;
            us_split8(y, h, l);
            if (vsl[r] >= 0) {
                unsigned char* p = lds + UK_A_OFF + vsl[r] * 16;
                *reinterpret_cast<h8*>(p) = h;
                *reinterpret_cast<h8*>(p + US_A_PLANE) = l;
            }
        }
        __syncthreads();
        const unsigned char* buf = lds + UK_A_OFF + abase;
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            const bool last = ca + 1 == nA && s == 6;
            const h8* wn = last ? wB : wA + (size_t)(ca * 7 + s + 1) * wstep;      // next k-step (the last phase-A step fetches the first phase-B step)
            load_w(wn, nh, nl);
            const unsigned char* ap = buf + atap[s];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const h8 ah = *reinterpret_cast<const h8*>(ap + m * 2 * US_SZ * 16);
                const h8 al = *reinterpret_cast<const h8*>(ap + m * 2 * US_SZ * 16 + US_A_PLANE);
                us_mfma_block<NB>(hi[m], lo[m], ah, al, bh, bl);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) { bh[nb] = nh[nb]; bl[nb] = nl[nb]; }
        }
    }
    if (nA == 0) __syncthreads();

    // ---- phase B
    for (int cb = 0; cb < nB; ++cb) {
#pragma unroll
        for (int tz = 0; tz < 2; ++tz) {
            const int sidx = cb * 2 + tz + 1;
            const h8* wn = wB + (size_t)(sidx < nB * 2 ? sidx : nB * 2 - 1) * wstep;   // (the last step re-reads itself)
            load_w(wn, nh, nl);
            const unsigned char* ap = lds + UK_B_OFF + cb * 2 * US_B_PLANE + bbase + tz * US_BZ * 16;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const h8 ah = *reinterpret_cast<const h8*>(ap + m * US_BZ * 16);
                const h8 al = *reinterpret_cast<const h8*>(ap + m * US_BZ * 16 + US_B_PLANE);
                us_mfma_block<NB>(hi[m], lo[m], ah, al, bh, bl);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) { bh[nb] = nh[nb]; bl[nb] = nl[nb]; }
        }
    }
    __syncthreads();

    // ---- epilogue: relu(hi + lo / 2^11) -> LDS tile [couts of this group][z][y][x] of the box -> float4 half rows, statistics of the box per cout
    float* e = reinterpret_cast<float*>(lds);
    {
        const int col = lane & 15, yj = lane >> 4;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lin = (2 * m + pz) * 64 + (2 * yj + py) * 8 + 2 * r + px;
                    e[(nb * 16 + col) * UB_E_STRIDE + lin] = fmaxf(fmaf(lo[m][nb][r], 1.0f / US_LO, hi[m][nb][r]), 0.f);
                }
    }
    __syncthreads();
    const int cob = nb0 * 16;
    int rows = a.cout - cob;
    if (rows > NB * 16) rows = NB * 16;
    float* __restrict__ o = a.out + ((size_t)n * a.cout + cob) * vol + ((size_t)z0 * edge + y0) * edge + x0;
    for (int q = tid; q < rows * 128; q += 512) {
        const int co = q >> 7, l4 = q & 127;
        const int z = l4 >> 4, y = (l4 >> 1) & 7, xh = l4 & 1;
        *reinterpret_cast<float4*>(o + (size_t)co * vol + ((size_t)z * edge + y) * edge + xh * 4) = *reinterpret_cast<const float4*>(e + co * UB_E_STRIDE + l4 * 4);
    }
    if (a.stats) {
        const int tiles = tpe * tpe * tpe;
        const int co = tid >> 3, part = tid & 7;
        double sm = 0.0, sq = 0.0;
        if (co < rows) {
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(e + co * UB_E_STRIDE + (part * 16 + i) * 4);
                sm += (double)v.x; sq += (double)v.x * v.x;
                sm += (double)v.y; sq += (double)v.y * v.y;
                sm += (double)v.z; sq += (double)v.z * v.z;
                sm += (double)v.w; sq += (double)v.w * v.w;
            }
        }
#pragma unroll
        for (int msk = 1; msk < 8; msk <<= 1) { sm += __shfl_xor(sm, msk, 64); sq += __shfl_xor(sq, msk, 64); }
        if (part == 0 && co < rows) a.stats[((size_t)n * a.cout + cob + co) * tiles + tile] = make_double2(sm, sq);
    }
}

// 8^3 boxes with a skip source: channel counts in eights, up to 12 low-res groups (LDS), enough boxes
static bool up_split_boxskip_takes(int c0, int c1, int n, int edge, int cout) {
    if (c0 < 8 || c0 % 8 || c1 <= 0 || c1 % 8 || c1 > 96 || cout <= 0 || cout > 96 || !rf_is_pow2(edge) || edge < 16 || edge > 128) return false;
    return (long long)n * (edge / 8) * (edge / 8) * (edge / 8) >= 512;
}

static bool up_split_box_takes(int c0, int c1, int n, int edge, int cout) {
    if (c0 != 0 || c1 <= 0 || c1 % 8 || c1 > 8 * US_MAX_CGB || cout <= 0 || cout > 32 || !rf_is_pow2(edge) || edge < 16 || edge > 128) return false;
    // >= 256 boxes: the launch need not fill the chip -- the chunk-level U-Net runs on a side stream beside the retrieval path -- but below that the
    // position-major fp32 kernels' latency wins (B = 32 chunks: the 16^3 stage is 256 boxes, 0.058 ms as k_conv3_up on the fp32 matrix path)
    return (long long)n * (edge / 8) * (edge / 8) * (edge / 8) >= 256;
}

extern "C" int rf_conv3d_up_split_stats_tiles(int c0, int c1, int n, int edge, int cout) {
    return (up_split_box_takes(c0, c1, n, edge, cout) || up_split_boxskip_takes(c0, c1, n, edge, cout)) ? (edge / 8) * (edge / 8) * (edge / 8) : 1;
}

extern "C" int rf_conv3d_up_split_supported(int c0, int c1, int n, int edge, int cout) {
    // whole 4^3 samples (k_conv3_up_split_s4): 8 per workgroup, 32 couts per workgroup
    if (edge == 4) return n >= 1024 && c0 >= 0 && c1 > 0 && c0 % 8 == 0 && c1 % 8 == 0 && cout > 0;
    if (up_split_box_takes(c0, c1, n, edge, cout)) return 1;      // box tiles of a large volume, upsampled channels only (k_conv3_up_split_box)
    if (up_split_boxskip_takes(c0, c1, n, edge, cout)) return 1;  // ... with a skip source (k_conv3_up_split_boxskip)
    if (edge != 8 || n < 256 || c0 < 0 || c1 <= 0 || c0 % 8 || c1 % 8 || c1 > 8 * US_MAX_CGB || cout <= 0) return 0;
    const int nb = rf_round_up(cout, 16) / 16;
    return nb == 3 || nb == 4;
}

template <int NB>
static int launch_up_split(const UpSplitArgs& a, hipStream_t stream) {
    auto kern = k_conv3_up_split<NB>;
    static RfLdsOptIn opt_in;
    if (int rc = opt_in.ensure(reinterpret_cast<const void*>(kern), US_LDS_ALLOC, "rf_conv3d_up_split_k3_gn_relu")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)a.n), dim3(512), US_LDS_ALLOC, stream, a);
    RF_CHECK_LAUNCH("rf_conv3d_up_split_k3_gn_relu");
    return RF_OK;
}

static int up_split_dispatch(UpSplitArgs& a, int c0, int c1, int n, int edge, int cout, void* stream);

// The persistent form (k_conv3_up_split_pp) takes the pre-split launches with four cout blocks, an even number (>= 4) of skip chunks and enough samples for
// every CU to walk several: one workgroup per CU (134 KB of LDS, 256 VGPRs), RF_UP_PP_ROUNDS rounds of them (see rf_persistent_wgs on why more than one).
#ifndef RF_UP_PP
#define RF_UP_PP 1
#endif
#ifndef RF_UP_PP_ROUNDS
#define RF_UP_PP_ROUNDS 1
#endif
#ifndef RF_UP_PP_LINEAR
#define RF_UP_PP_LINEAR 1                                          // development: 0 = the linear-order entry point stays on k_conv3_up_split
#endif
static bool up_split_pp_takes(int c0, int c1, int n, int cout) {
    return RF_UP_PP && cout > 48 && cout <= 64 && cout % 8 == 0 && c0 >= 32 && c0 % 16 == 0 && c1 >= 8 && c1 % 8 == 0 && c1 <= 8 * US_MAX_CGB && n >= 1024;
}

static int launch_up_split_pp(const UpSplitArgs& a, hipStream_t stream) {
    static RfLdsOptIn opt_in;
    if (int rc = opt_in.ensure(reinterpret_cast<const void*>(k_conv3_up_split_pp), PP_LDS_ALLOC, "rf_conv3d_up_split_presplit")) return rc;
    const int wgs = (rf_resident_wgs() / 2) * RF_UP_PP_ROUNDS;
    hipLaunchKernelGGL(k_conv3_up_split_pp, dim3((unsigned)(a.n < wgs ? a.n : wgs)), dim3(512), PP_LDS_ALLOC, stream, a);
    RF_CHECK_LAUNCH("rf_conv3d_up_split_presplit");
    return RF_OK;
}

extern "C" int rf_conv3d_up_split_k3_gn_relu(const float* src0, int c0, const float* src1, int c1, int n, int edge, const float* gn_affine,
                                              const void* w_packed, int cout, float* out, double* stats, void* stream) {
    RF_REQUIRE(rf_conv3d_up_split_supported(c0, c1, n, edge, cout), RF_E_UNSUPPORTED,
               "rf_conv3d_up_split_k3_gn_relu: takes whole 8^3 samples (n >= 256, c1 <= 64, 33..64 couts), 4^3 samples (n >= 1024) or 8^3 boxes of edge >= 16 volumes (without a skip source: c1 <= 64, <= 32 couts, >= 256 boxes; with one: c1 <= 96, <= 96 couts, >= 512 boxes); c0 and c1 in multiples of 8 (got c0=%d c1=%d n=%d edge=%d cout=%d)",
               c0, c1, n, edge, cout);
    RF_REQUIRE((c0 == 0 || src0) && src1 && gn_affine && w_packed && out, RF_E_INVALID, "rf_conv3d_up_split_k3_gn_relu: null pointer");
    UpSplitArgs a;
    a.src0 = src0; a.src1 = src1; a.affine = reinterpret_cast<const float4*>(gn_affine); a.wp = reinterpret_cast<const h8*>(w_packed);
    a.out = out; a.stats = reinterpret_cast<double2*>(stats); a.c0 = c0; a.c1 = c1; a.n = n; a.cout = cout;
    a.pre_out = nullptr; a.ngamma = a.nbeta = nullptr; a.ngroups = 0; a.neps = 0.f; a.out_ch8 = 0; a.pre_pm = 0;
    return up_split_dispatch(a, c0, c1, n, edge, cout, stream);
}

static int up_split_dispatch(UpSplitArgs& a, int c0, int c1, int n, int edge, int cout, void* stream) {
    if (up_split_box_takes(c0, c1, n, edge, cout)) {
        const unsigned boxes = (unsigned)n * (edge / 8) * (edge / 8) * (edge / 8);
        const int nbq = rf_round_up(cout, 16) / 16;
        const size_t lds_bytes = (size_t)(nbq * 16) * UB_E_STRIDE * 4 > (size_t)(c1 / 8) * 2 * US_B_PLANE ? (size_t)(nbq * 16) * UB_E_STRIDE * 4 : (size_t)(c1 / 8) * 2 * US_B_PLANE;
        if (a.out_ch8 && (c1 == 8 || c1 == 16) && (cout == 8 || cout == 16) && boxes >= 2048 && (long long)cout * edge * edge * edge < (1ll << 28)) {
            // persistent form (channel-interleaved output only): two workgroups per CU, each a run of consecutive boxes
            static RfLdsOptIn opt_p[4];
#define RF_BOXP(I_, NBG_, CGO_)                                                                                                      \
            do {                                                                                                                     \
                if (int rc = opt_p[I_].ensure(reinterpret_cast<const void*>(k_conv3_up_split_boxp<NBG_, CGO_>), UP_LDS_BYTES, "rf_conv3d_up_split_k3_gn_relu_ch8")) return rc; \
                hipLaunchKernelGGL((k_conv3_up_split_boxp<NBG_, CGO_>), dim3((unsigned)rf_persistent_wgs()), dim3(512), UP_LDS_BYTES, (hipStream_t)stream, a, edge, (int)boxes); \
            } while (0)
            if (c1 == 16 && cout == 16) RF_BOXP(0, 2, 2);
            else if (c1 == 16) RF_BOXP(1, 2, 1);
            else if (cout == 16) RF_BOXP(2, 1, 2);
            else RF_BOXP(3, 1, 1);
#undef RF_BOXP
        } else if (nbq == 1) {
            hipLaunchKernelGGL(k_conv3_up_split_box<1>, dim3(boxes), dim3(512), lds_bytes, (hipStream_t)stream, a, edge);
        } else {
            static RfLdsOptIn opt_in;
            if (int rc = opt_in.ensure(reinterpret_cast<const void*>(k_conv3_up_split_box<2>), (int)lds_bytes, "rf_conv3d_up_split_k3_gn_relu")) return rc;
            hipLaunchKernelGGL(k_conv3_up_split_box<2>, dim3(boxes), dim3(512), lds_bytes, (hipStream_t)stream, a, edge);
        }
        RF_CHECK_LAUNCH("rf_conv3d_up_split_k3_gn_relu");
        return RF_OK;
    }
    if (up_split_boxskip_takes(c0, c1, n, edge, cout)) {
        const unsigned boxes = (unsigned)n * (edge / 8) * (edge / 8) * (edge / 8);
        const int nbt = rf_round_up(cout, 16) / 16;
        const size_t img = (size_t)US_A_BUF + (size_t)(c1 / 8) * 2 * US_B_PLANE;
        static RfLdsOptIn opt3, opt2, opt1;
#define RF_BOXSKIP(NB_, OPT_)                                                                                                        \
        do {                                                                                                                         \
            const size_t tile = (size_t)(NB_ * 16) * UB_E_STRIDE * 4;                                                                \
            const size_t lds_bytes = tile > img ? tile : img;                                                                        \
            if (int rc = OPT_.ensure(reinterpret_cast<const void*>(k_conv3_up_split_boxskip<NB_>), 160 * 1024, "rf_conv3d_up_split_k3_gn_relu")) return rc; \
            hipLaunchKernelGGL(k_conv3_up_split_boxskip<NB_>, dim3(boxes, (unsigned)((nbt + NB_ - 1) / NB_)), dim3(512), lds_bytes, (hipStream_t)stream, a, edge); \
        } while (0)
        if (nbt == 1) RF_BOXSKIP(1, opt1);
        else if (nbt == 2 || nbt == 4) RF_BOXSKIP(2, opt2);
        else RF_BOXSKIP(3, opt3);
#undef RF_BOXSKIP
        RF_CHECK_LAUNCH("rf_conv3d_up_split_k3_gn_relu");
        return RF_OK;
    }
    if (edge == 4) {
        const int nbt = rf_round_up(cout, 16) / 16;
        if (US_S4_WIDE && (nbt == 4 || nbt == 3)) {
            // all 64 (48: nf = 12) couts in one workgroup (256 VGPRs, 132 KB LDS, one workgroup per CU -- the shape of the 8^3 kernel): the samples are
            // staged and converted ONCE instead of once per 16-cout block
            static RfLdsOptIn opt_wide4, opt_wide3;
            const int lds_wide = nbt * 16 * U4_E_STRIDE * 4;
            if (nbt == 4) {
                if (int rc = opt_wide4.ensure(reinterpret_cast<const void*>(k_conv3_up_split_s4<4, true>), lds_wide, "rf_conv3d_up_split_k3_gn_relu")) return rc;
                hipLaunchKernelGGL((k_conv3_up_split_s4<4, true>), dim3((unsigned)((n + 7) / 8), 1u), dim3(512), lds_wide, (hipStream_t)stream, a);
            } else {
                if (int rc = opt_wide3.ensure(reinterpret_cast<const void*>(k_conv3_up_split_s4<3, true>), lds_wide, "rf_conv3d_up_split_k3_gn_relu")) return rc;
                hipLaunchKernelGGL((k_conv3_up_split_s4<3, true>), dim3((unsigned)((n + 7) / 8), 1u), dim3(512), lds_wide, (hipStream_t)stream, a);
            }
            RF_CHECK_LAUNCH("rf_conv3d_up_split_k3_gn_relu");
            return RF_OK;
        }
        static RfLdsOptIn opt_in;
        // 16 couts per workgroup: the 32-cout instance needs more than the 128 VGPRs that four waves per SIMD allow (35-45 spills) and was
        // no faster (629-641 us against 610 on dec0)
        if (int rc = opt_in.ensure(reinterpret_cast<const void*>(k_conv3_up_split_s4<1, true>), U4_LDS_BYTES, "rf_conv3d_up_split_k3_gn_relu")) return rc;
        hipLaunchKernelGGL((k_conv3_up_split_s4<1, true>), dim3((unsigned)((n + 7) / 8), (unsigned)nbt), dim3(512), U4_LDS_BYTES,
                           (hipStream_t)stream, a);
        RF_CHECK_LAUNCH("rf_conv3d_up_split_k3_gn_relu");
        return RF_OK;
    }
    return rf_round_up(cout, 16) == 48 ? launch_up_split<3>(a, (hipStream_t)stream) : launch_up_split<4>(a, (hipStream_t)stream);
}

// The whole-sample decoder form with its output handed to the NEXT SingleConv pre-split: relu(conv(GN(x))) of 8^3 samples, then the
// next layer's GroupNorm (next_gamma / next_beta [cout], next_groups, eps) applied from the sample's own statistics, scaled and split into f16 pairs:
// out_presplit = rf_split_act_bytes(n, cout, 8) bytes for rf_conv3d_split_pre_k3_relu.  stats (optional) as rf_conv3d_up_split_k3_gn_relu.
extern "C" int rf_conv3d_up_split_presplit_supported(int c0, int c1, int n, int edge, int cout, int next_groups) {
    return edge == 8 && rf_conv3d_up_split_supported(c0, c1, n, edge, cout) && !up_split_box_takes(c0, c1, n, edge, cout) && cout % 8 == 0 && next_groups > 0 &&
           cout % next_groups == 0 && cout <= 64;
}

extern "C" int rf_conv3d_up_split_presplit(const float* src0, int c0, const float* src1, int c1, int n, int edge, const float* gn_affine, const void* w_packed,
                                           int cout, const float* next_gamma, const float* next_beta, int next_groups, float eps, void* out_presplit,
                                           double* stats, void* stream) {
    RF_REQUIRE(rf_conv3d_up_split_presplit_supported(c0, c1, n, edge, cout, next_groups), RF_E_UNSUPPORTED,
               "rf_conv3d_up_split_presplit: takes whole 8^3 samples (the shapes rf_conv3d_up_split_supported takes at edge 8) with cout in eights and in whole groups (got c0=%d c1=%d n=%d edge=%d cout=%d groups=%d)",
               c0, c1, n, edge, cout, next_groups);
    RF_REQUIRE((c0 == 0 || src0) && src1 && gn_affine && w_packed && out_presplit && next_gamma && next_beta, RF_E_INVALID, "rf_conv3d_up_split_presplit: null pointer");
    UpSplitArgs a;
    a.src0 = src0; a.src1 = src1; a.affine = reinterpret_cast<const float4*>(gn_affine); a.wp = reinterpret_cast<const h8*>(w_packed);
    a.out = nullptr; a.stats = reinterpret_cast<double2*>(stats); a.c0 = c0; a.c1 = c1; a.n = n; a.cout = cout;
    a.pre_out = reinterpret_cast<h8*>(out_presplit); a.ngamma = next_gamma; a.nbeta = next_beta; a.ngroups = next_groups; a.neps = eps; a.out_ch8 = 0; a.pre_pm = 0;
    if (RF_UP_PP_LINEAR && up_split_pp_takes(c0, c1, n, cout)) return launch_up_split_pp(a, (hipStream_t)stream);
    return rf_round_up(cout, 16) == 48 ? launch_up_split<3>(a, (hipStream_t)stream) : launch_up_split<4>(a, (hipStream_t)stream);
}

// The same with the voxel slots of the output in PARITY-MAJOR order (UpSplitArgs::pre_pm; consumer: rf_conv3d_split_pre_pm_k3_relu): the persistent kernel
// k_conv3_up_split_pp, whose waves own one output parity each and leave their slots from registers -- in the linear order those are 16-byte pieces 32 bytes
// apart, in this order 256-byte runs.
extern "C" int rf_conv3d_up_split_presplit_pm_supported(int c0, int c1, int n, int edge, int cout, int next_groups) {
    return rf_conv3d_up_split_presplit_supported(c0, c1, n, edge, cout, next_groups) && up_split_pp_takes(c0, c1, n, cout);
}

extern "C" int rf_conv3d_up_split_presplit_pm(const float* src0, int c0, const float* src1, int c1, int n, int edge, const float* gn_affine, const void* w_packed,
                                              int cout, const float* next_gamma, const float* next_beta, int next_groups, float eps, void* out_presplit_pm,
                                              double* stats, void* stream) {
    RF_REQUIRE(rf_conv3d_up_split_presplit_pm_supported(c0, c1, n, edge, cout, next_groups), RF_E_UNSUPPORTED,
               "rf_conv3d_up_split_presplit_pm: takes whole 8^3 samples (n >= 1024), 49..64 couts in eights and in whole groups, c0 >= 32 in sixteens, c1 <= 64 in eights (got c0=%d c1=%d n=%d edge=%d cout=%d groups=%d)",
               c0, c1, n, edge, cout, next_groups);
    RF_REQUIRE(src0 && src1 && gn_affine && w_packed && out_presplit_pm && next_gamma && next_beta, RF_E_INVALID, "rf_conv3d_up_split_presplit_pm: null pointer");
    UpSplitArgs a;
    a.src0 = src0; a.src1 = src1; a.affine = reinterpret_cast<const float4*>(gn_affine); a.wp = reinterpret_cast<const h8*>(w_packed);
    a.out = nullptr; a.stats = reinterpret_cast<double2*>(stats); a.c0 = c0; a.c1 = c1; a.n = n; a.cout = cout;
    a.pre_out = reinterpret_cast<h8*>(out_presplit_pm); a.ngamma = next_gamma; a.nbeta = next_beta; a.ngroups = next_groups; a.neps = eps; a.out_ch8 = 0; a.pre_pm = 1;
    return launch_up_split_pp(a, (hipStream_t)stream);
}

// rf_conv3d_up_split_k3_gn_relu with the output CHANNEL-INTERLEAVED, [n][cout / 8][edge^3][8 channels] fp32 ("ch8"), for a consumer that stages the 8 channels
// of a voxel together (rf_conv3d_split_k3_gn_relu_pointwise_tanh_ch8: the final decoder's conv pair on 64^3, reference model/refinement.py:48-61 -- a
// 64^3 sample is 512 boxes, no workgroup has its statistics, so the pair cannot hand over pre-split; but the consumer's staging of an NCDHW tensor is
// 16 four-byte gathers per thread and chunk in 40-byte runs, bound by the address path, where this layout gives it four 16-byte loads).  Same values,
// same statistics; the box form only (no skip source, <= 32 couts in eights).
extern "C" int rf_conv3d_up_split_ch8_supported(int c0, int c1, int n, int edge, int cout) {
    return up_split_box_takes(c0, c1, n, edge, cout) && cout % 8 == 0;
}

extern "C" int rf_conv3d_up_split_k3_gn_relu_ch8(const float* src0, int c0, const float* src1, int c1, int n, int edge, const float* gn_affine,
                                                  const void* w_packed, int cout, float* out_ch8, double* stats, void* stream) {
    RF_REQUIRE(rf_conv3d_up_split_ch8_supported(c0, c1, n, edge, cout), RF_E_UNSUPPORTED,
               "rf_conv3d_up_split_k3_gn_relu_ch8: takes the box form of rf_conv3d_up_split_k3_gn_relu (no skip source, >= 256 boxes) with cout in eights (got c0=%d c1=%d n=%d edge=%d cout=%d)",
               c0, c1, n, edge, cout);
    RF_REQUIRE(src1 && gn_affine && w_packed && out_ch8, RF_E_INVALID, "rf_conv3d_up_split_k3_gn_relu_ch8: null pointer");
    UpSplitArgs a;
    a.src0 = src0; a.src1 = src1; a.affine = reinterpret_cast<const float4*>(gn_affine); a.wp = reinterpret_cast<const h8*>(w_packed);
    a.out = out_ch8; a.stats = reinterpret_cast<double2*>(stats); a.c0 = c0; a.c1 = c1; a.n = n; a.cout = cout;
    a.pre_out = nullptr; a.ngamma = a.nbeta = nullptr; a.ngroups = 0; a.neps = 0.f; a.out_ch8 = 1; a.pre_pm = 0;
    return up_split_dispatch(a, c0, c1, n, edge, cout, stream);
}
