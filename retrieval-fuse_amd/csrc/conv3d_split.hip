// rf_conv3d_split_k3_gn_relu: SingleConv 'gcr' (reference model/unet.py:19-76: GroupNorm -> 3x3x3 conv, pad 1, no bias -> ReLU) on 8^3
// output boxes -- the arithmetic contract and the outputs (ReLU'd volume, GroupNorm statistics, fused MaxPool3d(2)) of
// rf_conv3d_k3_gn_relu[_stats|_pool] (conv3d_mfma.hip), evaluated on the F16 matrix cores by OPERAND SPLITTING:
//     x = h + l / 2^11,  h = f16(x),  l = f16((x - h) * 2^11);   a*b ~ ah*bh + (ah*bl + al*bh) / 2^11,   exact f16 x f16 products,
//     fp32 accumulation in two separate accumulators (hi, lo), combined once in the epilogue
// (see conv3d_up_split.hip for the numerics: half the rounding error of the fp32 MFMA chain at 5.3x its multiply-add rate).
//
// GEMM view: M = the 512 voxels of a box (8 waves x 4 m-blocks, voxel order = BoxOrder of conv_box.h so that the epilogue -- shared with
// the fp32 kernel -- finds whole float4 rows and pooling cells in a lane), N = cout (NB = 1 or 2 n-blocks), K = cin * 27 walked in
// chunks of 8 input channels, 7 k-steps each: an MFMA k-step (k = 32) is 4 TAPS x 8 CHANNELS -- lane group g = lane >> 4 supplies tap
// 4s + g (tap 27 is a zero-weight dummy), its 8 halves are the chunk's 8 channels of one voxel.  The LDS image of a chunk is the halo
// box [10][10][10] in 16-byte slots (8 channels of a voxel), one plane for h and one for l, so an A operand is one ds_read_b128 at
// (voxel + tap offset).  Two chunk buffers: the next chunk's raw voxels (two halo voxels per thread, 8 channels each) are requested
// before the MFMAs of the current chunk and normalised + split + written to the other buffer after them; one barrier per chunk.
// B operands (weights): f16 fragment image from rf_conv3_split_pack_weight ([chunk][k-step][n-block][h|l][lane][8 halves]), L2-resident,
// global -> VGPR one k-step ahead.  Register use is small (NB 1: 32 accumulator VGPRs), two workgroups per CU.
#include "common.h"
#include "conv_box.h"
#include "conv_split_common.h"
#include <type_traits>

// conv3d_split_zc.hip: the persistent z-column forms.  zc: the one-chunk pre-split layer on 16^3 samples (8 -> 16: level 0 of the retrieval backbone);
// zcm: layers of two or more whole chunks with up to 16 couts, any output form of this file (full / pooled / pointwise head / pre-split)
bool rf_split_zc_takes(int cin, int n, int edge, int cout);
int rf_split_zc_launch(const ConvArgs& a, const SplitPreOut& po, hipStream_t stream);
bool rf_split_zcm_takes(int cin, int n, int edge, int cout);
int rf_split_zcm_launch(const ConvArgs& a, const SplitPreOut& po, bool pre, hipStream_t stream, const char* who);
int rf_split_zcm_launch_ch8_pointwise(const ConvArgs& a, const SplitPreOut& po, hipStream_t stream, const char* who);

// ------------------------------------------------------------------------------------------------------------ weight image
extern "C" size_t rf_conv3_split_packed_bytes(int cout, int cin) {
    return ((size_t)((cin + 7) / 8) * 7 + 1) * (size_t)(rf_round_up(cout, 16) / 16) * 2 * 64 * 16;
}

__global__ void k_conv3_split_pack(const float* __restrict__ w, int cout, int cin, int nb_count, h8* __restrict__ wp, size_t total) {
    const size_t nreal = (size_t)((cin + 7) / 8) * 7 * nb_count * 128;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), piece = (int)((i >> 6) & 1);
        const size_t f = i >> 7;
        const int nb = (int)(f % nb_count);
        const size_t st = f / nb_count;
        const int co = nb * 16 + (lane & 15), g = lane >> 4;
        const int s = (int)(st % 7), ca = (int)(st / 7), tap = 4 * s + g;
        h8 out;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double v = 0.0;
            if (i < nreal && tap < 27 && co < cout && ca * 8 + j < cin) v = (double)w[((size_t)co * cin + ca * 8 + j) * 27 + tap];   // channels past cin: zero
            v *= (double)CS_W_SCALE;
            v = v > 65504.0 ? 65504.0 : (v < -65504.0 ? -65504.0 : v);
            const _Float16 h = (_Float16)(float)v;
            out[j] = piece == 0 ? h : (_Float16)(float)((v - (double)(float)h) * (double)CS_LO);
        }
        wp[i] = out;
    }
}

extern "C" int rf_conv3_split_pack_weight(const float* w_oidhw, int cout, int cin, void* w_packed, void* stream) {
    RF_REQUIRE(w_oidhw && w_packed && cout > 0 && cin > 0, RF_E_INVALID, "rf_conv3_split_pack_weight: bad arguments");
    const size_t total = rf_conv3_split_packed_bytes(cout, cin) / 16;
    const size_t want = (total + 255) / 256;
    hipLaunchKernelGGL(k_conv3_split_pack, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream, w_oidhw, cout, cin,
                       rf_round_up(cout, 16) / 16, reinterpret_cast<h8*>(w_packed), total);
    RF_CHECK_LAUNCH("rf_conv3_split_pack_weight");
    return RF_OK;
}

// ------------------------------------------------------------------------------------------------------------------- kernel
// ONE: the layer has a single 8-channel chunk (no prefetch of a next chunk, one chunk buffer)
// PADC: cin is not a multiple of 8 -- the last chunk's missing channels are staged as zeros (they re-read channel cin-1 with a zero affine
// triple; their weights are zero too)
// PRE: the input is a PRE-SPLIT tensor (rf_split_act_bytes: per (sample, 8-channel group) an h plane and an l plane of 16-byte voxel slots) that
// its producer already normalised for THIS layer's GroupNorm and split -- staging is a copy of slots, no affine table, no conversion
constexpr int CS_PO_STRIDE = 517;                                   // tile row (floats), odd: conflict-free scalar writes
constexpr int CS_PO_STATS = 16 * CS_PO_STRIDE * 4, CS_PO_TRIPLES = CS_PO_STATS + 16 * 16;      // behind the tile: 16 x double2, 16 x float4
static_assert(CS_PO_TRIPLES + 16 * 16 <= CS_LDS_BYTES, "pre-split epilogue must fit the multi-chunk instance's LDS");

template <int NB, int WPS, bool ONE, bool PADC, bool PRE = false>
__global__ __launch_bounds__(512, WPS) void k_conv3_split(ConvArgs a, SplitPreOut po) {
    static_assert(!(PRE && PADC), "pre-split tensors carry whole 8-channel groups");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int edge = a.edge, cin = a.c0, nC = (cin + 7) >> 3;
    const size_t vol = (size_t)edge * edge * edge;

    const unsigned lblock = rf_xcd_contiguous(blockIdx.x, gridDim.x);
    const int tpe = edge >> 3;
    int t = (int)lblock;
    const int x0 = (t % tpe) * 8; t /= tpe;
    const int y0 = (t % tpe) * 8; t /= tpe;
    const int z0 = (t % tpe) * 8; t /= tpe;
    const int n0 = t;
    const int cob = blockIdx.y * (NB * 16);
    const float4* __restrict__ aff = a.affine + (size_t)n0 * cin;
    auto chan = [&](int c) { return PADC ? (c < cin ? c : cin - 1) : c; };                     // uniform
    auto triple = [&](int c) { return (!PADC || c < cin) ? aff[c] : make_float4(0.f, 0.f, 0.f, 0.f); };

    // ---- staging: thread t owns halo voxels t and t + 512 (the second only for t < 488)
    int voff[2], vslot[2];
    bool vin[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int v = tid + r * 512;
        const int hx = v % 10, hy = (v / 10) % 10, hz = v / 100;
        const int z = z0 + hz - 1, y = y0 + hy - 1, x = x0 + hx - 1;
        vin[r] = v < CS_VOX && (unsigned)z < (unsigned)edge && (unsigned)y < (unsigned)edge && (unsigned)x < (unsigned)edge;
        voff[r] = vin[r] ? (z * edge + y) * edge + x : 0;
        vslot[r] = (hz * CS_SZ + hy * CS_SY + hx) * 16;              // byte offset of the voxel's slot in a plane
    }
    const float* __restrict__ s0 = a.src0 + (size_t)n0 * cin * vol;
    const unsigned char* __restrict__ sp0 = reinterpret_cast<const unsigned char*>(a.src0) + (size_t)n0 * nC * 2 * vol * 16;      // PRE
    struct Staged { float x[PRE ? 1 : 2][PRE ? 1 : 8]; h8 ph[PRE ? 2 : 1], pl[PRE ? 2 : 1]; };
    auto stage_load = [&](Staged& st, int ca) {
        if constexpr (PRE) {
            const unsigned char* p = sp0 + (size_t)ca * 2 * vol * 16;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                st.ph[r] = *reinterpret_cast<const h8*>(p + (size_t)voff[r] * 16);
                st.pl[r] = *reinterpret_cast<const h8*>(p + (vol + (size_t)voff[r]) * 16);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) st.x[r][j] = s0[(size_t)chan(ca * 8 + j) * vol + voff[r]];
        }
    };
    auto stage_store = [&](const Staged& st, int ca, int buf) {      // zeros outside the volume (the padding of the NORMALISED tensor)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            h8 h, l;
            if constexpr (PRE) {
                const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
                h = vin[r] ? st.ph[r] : zero;
                l = vin[r] ? st.pl[r] : zero;
            } else {
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 af = triple(ca * 8 + j);
                    y[j] = vin[r] ? fmaf(st.x[r][j] - af.x, af.y, af.z) : 0.f;
                }
                cs_split8(y, h, l);
            }
            if (r == 0 || tid < CS_VOX - 512) {
                unsigned char* p = lds + buf * CS_BUF + vslot[r];
                *reinterpret_cast<h8*>(p) = h;
                *reinterpret_cast<h8*>(p + CS_PLANE) = l;
            }
        }
    };
    // the first chunk's voxels are requested now and staged after the addressing set-up and the first weight fragments are under way
    // (measured with s_memtime: set-up + first weight load cost a quarter of a workgroup's life when they came after the staging)
    Staged xs0;
    stage_load(xs0, 0);
    __builtin_amdgcn_sched_barrier(0);

    // ---- per-lane operand addressing (BoxOrder: x = i & 7, y = 4 (wave & 1) + 2 (mb >> 1) + (i >> 3), z = 2 (wave >> 1) + (mb & 1))
    const int g = lane >> 4, ri = lane & 15;
    const int abase = ((2 * (wave >> 1) + 1) * CS_SZ + (4 * (wave & 1) + (ri >> 3) + 1) * CS_SY + (ri & 7) + 1) * 16;
    // tap offset of k-step s for this lane's tap 4 s + g.  Multi-chunk instances keep the seven offsets in registers; the one-chunk instances
    // (80 registers for three workgroups per CU; the seven offsets were exactly the seven registers they spilled, reloaded from scratch between the
    // MFMAs of the k-steps) recompute an offset in the shadow of the previous k-step's MFMAs, from a copy of g the compiler cannot see through
    int atap[ONE ? 1 : 7];
    auto tap_off = [&](int s) -> int {
        if constexpr (ONE) {
            int gg = g;
            asm volatile("" : "+v"(gg));
            const int tq = 4 * s + gg, tp = tq < 27 ? tq : 26;
            return ((tp / 9 - 1) * CS_SZ + ((tp / 3) % 3 - 1) * CS_SY + (tp % 3 - 1)) * 16;
        } else {
            return atap[s];
        }
    };
    if constexpr (!ONE) {
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            const int tp = 4 * s + g < 27 ? 4 * s + g : 26;
            atap[s] = ((tp / 9 - 1) * CS_SZ + ((tp / 3) % 3 - 1) * CS_SY + (tp % 3 - 1)) * 16;
        }
    }

    f32x4 hi[4][NB], lo[4][NB];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { hi[m][nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; lo[m][nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const int nbt = a.cout16 >> 4;                                        // n-blocks of the weight image
    const h8* __restrict__ wn = reinterpret_cast<const h8*>(a.wp) + (size_t)blockIdx.y * NB * 128 + lane;      // next k-step to fetch
    const int wstep = nbt * 128;
    h8 b0h[NB], b0l[NB], b1h[NB], b1l[NB];
    auto load_b = [&](h8 (&bh)[NB], h8 (&bl)[NB]) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            bh[nb] = wn[(nb * 2) * 64];
            bl[nb] = wn[(nb * 2 + 1) * 64];
        }
        wn += wstep;
    };
    load_b(b0h, b0l);
    __builtin_amdgcn_sched_barrier(0);
    stage_store(xs0, 0, 0);
    h8 ah[2], al[2];
    __syncthreads();

    // one k-step (see conv3d_up_split.hip): next step's B fragments first, then (first step of a chunk) the next chunk's raw voxels BEHIND
    // them (vmcnt retires in order), A operands of m-block m+1 under the MFMAs of m-block m; sched_barriers pin the order
    auto kstep = [&](auto has_pre, auto&& xload, auto&& hook, const unsigned char* ap, const unsigned char* pre, const h8 (&bh)[NB], const h8 (&bl)[NB], h8 (&nh)[NB], h8 (&nl)[NB]) {
        load_b(nh, nl);
        xload();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            constexpr int MOFF[4] = {0, CS_SZ * 16, 2 * CS_SY * 16, (CS_SZ + 2 * CS_SY) * 16};
            if (m < 3) {
                ah[(m + 1) & 1] = *reinterpret_cast<const h8*>(ap + MOFF[(m + 1) & 3]);
                al[(m + 1) & 1] = *reinterpret_cast<const h8*>(ap + MOFF[(m + 1) & 3] + CS_PLANE);
            } else if constexpr (decltype(has_pre)::value) {
                ah[0] = *reinterpret_cast<const h8*>(pre);
                al[0] = *reinterpret_cast<const h8*>(pre + CS_PLANE);
            }
            __builtin_amdgcn_sched_barrier(0);
            hook(m);                                                   // VALU work of the next chunk's staging, issued in the shadow of this block's MFMAs
            cs_mfma_block<NB>(hi[m], lo[m], ah[m & 1], al[m & 1], bh, bl);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto no_x = [] {};

    // no branch inside a chunk (hipcc's s_waitcnt insertion assumes the worst at every join): the last chunk re-loads its own channels
    // and stages them into the idle buffer
    // The staging of chunk c+1 is VALU work (normalise, split: ~7 instructions per value, 16 values per thread) as long as the MFMAs of a
    // chunk when it runs on its own (s_memtime: 11 k ticks against 8 k).  The loads are issued at k-step 0 and have landed by k-step 2
    // (vmcnt retires in order and k-step 2 waits for its weight fragments); k-steps 2..5 convert one value per m-block in the shadow of that
    // block's MFMAs; after k-step 6 only the four ds_write_b128 are left.
    auto chunk = [&](int ca, h8 (&ch)[NB], h8 (&cl)[NB], h8 (&nh)[NB], h8 (&nl)[NB]) {
        Staged x;
        h8 hq[ONE ? 1 : 2], lq[ONE ? 1 : 2];
        const int cx = ca + 1 < nC ? ca + 1 : ca;
        float4 afn[(ONE || PRE) ? 1 : 8];
        if constexpr (!ONE && !PRE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) afn[j] = triple(cx * 8 + j);      // uniform: scalar loads, before the k-steps' LDS traffic
        }
        auto xload = [&] { if constexpr (!ONE) stage_load(x, cx); };
        auto no_hook = [](int) {};
        const unsigned char* buf = lds + (ca & 1) * CS_BUF + abase;
        int tcur = tap_off(0);
        ah[0] = *reinterpret_cast<const h8*>(buf + tcur);
        al[0] = *reinterpret_cast<const h8*>(buf + tcur + CS_PLANE);
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            const unsigned char* ap = buf + tcur;
            const int tnext = s < 6 ? tap_off(s + 1) : tcur;
            auto conv_hook = [&](int m) {
                if constexpr (!ONE && !PRE) {
                    const int e = (s - 2) * 4 + m, r = e >> 3, j = e & 7;
                    const float y = vin[r] ? fmaf(x.x[r][j] - afn[j].x, afn[j].y, afn[j].z) : 0.f;
                    const float v = __builtin_amdgcn_fmed3f(y * CS_ACT_SCALE, -65504.f, 65504.f);
                    const _Float16 hh = (_Float16)v;
                    hq[r][j] = hh;
                    lq[r][j] = (_Float16)fmaf(-CS_LO, (float)hh, v * CS_LO);
                }
            };
            if (s == 0) kstep(std::true_type{}, xload, no_hook, ap, buf + tnext, ch, cl, nh, nl);
            else if (s == 6) kstep(std::false_type{}, no_x, no_hook, ap, ap, ch, cl, nh, nl);
            else if (s == 1) kstep(std::true_type{}, no_x, no_hook, ap, buf + tnext, nh, nl, ch, cl);
            else if (s & 1) kstep(std::true_type{}, no_x, conv_hook, ap, buf + tnext, nh, nl, ch, cl);
            else kstep(std::true_type{}, no_x, conv_hook, ap, buf + tnext, ch, cl, nh, nl);
            tcur = tnext;
        }
        if constexpr (!ONE) {
            if constexpr (PRE) {
                const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int r = 0; r < 2; ++r) { hq[r] = vin[r] ? x.ph[r] : zero; lq[r] = vin[r] ? x.pl[r] : zero; }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (r == 0 || tid < CS_VOX - 512) {
                    unsigned char* p = lds + ((ca + 1) & 1) * CS_BUF + vslot[r];
                    *reinterpret_cast<h8*>(p) = hq[r];
                    *reinterpret_cast<h8*>(p + CS_PLANE) = lq[r];
                }
        }
        __syncthreads();
    };
    for (int ca = 0; ca < nC; ca += 2) {
        chunk(ca, b0h, b0l, b1h, b1l);
        if (ca + 1 < nC) chunk(ca + 1, b1h, b1l, b0h, b0l);
    }

    if constexpr (NB == 1 && !ONE) {
        if (po.out) {
            // ---- pre-split output: ReLU'd tile [16 couts][8^3] -> statistics of the sample -> the next layer's triples -> normalise, split, slots
            __syncthreads();                                           // the chunk images are dead
            float* e = reinterpret_cast<float*>(lds);
            {
                const int col = lane & 15;
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int sdummy, z, y, x;
                        BoxOrder<8, 8, 8, 8, 4>::voxel(wave, m, 4 * (lane >> 4) + r, sdummy, z, y, x);
                        e[col * CS_PO_STRIDE + (z * 8 + y) * 8 + x] = fmaxf(fmaf(lo[m][0][r], 1.0f / CS_LO, hi[m][0][r]), a.floor);
                    }
            }
            __syncthreads();
            double2* chst = reinterpret_cast<double2*>(lds + CS_PO_STATS);
            float4* trip = reinterpret_cast<float4*>(lds + CS_PO_TRIPLES);
            {
                const int co = tid >> 5, part = tid & 31;               // 32 threads per cout, 16 values each, then a butterfly (fixed order)
                double sm = 0.0, sq = 0.0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float v = e[co * CS_PO_STRIDE + part + 32 * i];
                    sm += (double)v; sq += (double)v * v;
                }
#pragma unroll
                for (int msk = 1; msk < 32; msk <<= 1) { sm += __shfl_xor(sm, msk, 64); sq += __shfl_xor(sq, msk, 64); }
                if (part == 0) {
                    chst[co] = make_double2(sm, sq);
                    if (a.stats && cob + co < a.cout) a.stats[(size_t)n0 * a.cout + cob + co] = make_double2(sm, sq);
                }
            }
            __syncthreads();
            if (tid < a.cout) {                                        // as rf_gn_from_stats: group sums in channel order, float64
                const int cpg = a.cout / po.groups, ca = (tid / cpg) * cpg;
                double sm = 0.0, sq = 0.0;
                for (int c = ca; c < ca + cpg; ++c) { sm += chst[c].x; sq += chst[c].y; }
                const double count = (double)cpg * 512.0, mean = sm / count;
                double var = sq / count - mean * mean;
                if (var < 0.0) var = 0.0;
                trip[tid] = gn_affine(mean, 1.0 / sqrt(var + (double)po.eps), po.gamma[tid], po.beta[tid]);
            }
            __syncthreads();
            h8* __restrict__ o = po.out + (size_t)n0 * (a.cout >> 3) * 2 * 512 + tid;
            for (int sg = 0; sg < (a.cout >> 3); ++sg) {
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 t4 = trip[sg * 8 + j];
                    y[j] = fmaf(e[(sg * 8 + j) * CS_PO_STRIDE + tid] - t4.x, t4.y, t4.z);
                }
                h8 h, l;
                cs_split8(y, h, l);
                o[(size_t)sg * 2 * 512] = h;
                o[(size_t)sg * 2 * 512 + 512] = l;
            }
            return;
        }
        if (po.pw_out) {
            // ---- pointwise head: ReLU'd tile [couts][8^3] -> per voxel the channel sum in the order of rf_conv1x1_tanh (bias first, channels ascending: same bits)
            __syncthreads();
            float* e = reinterpret_cast<float*>(lds);
            {
                const int col = lane & 15;
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int sdummy, z, y, x;
                        BoxOrder<8, 8, 8, 8, 4>::voxel(wave, m, 4 * (lane >> 4) + r, sdummy, z, y, x);
                        e[col * CS_PO_STRIDE + (z * 8 + y) * 8 + x] = fmaxf(fmaf(lo[m][0][r], 1.0f / CS_LO, hi[m][0][r]), a.floor);
                    }
            }
            __syncthreads();
            float accp = po.pw_b[0];
            for (int c = 0; c < a.cout; ++c) accp = fmaf(e[c * CS_PO_STRIDE + tid], po.pw_w[c], accp);
            const int z = tid >> 6, y = (tid >> 3) & 7, x = tid & 7;
            po.pw_out[(size_t)n0 * vol + ((size_t)(z0 + z) * edge + (y0 + y)) * edge + x0 + x] = (tanhf(accp) + po.post_add) * po.post_mul;
            return;
        }
    }
    // ---- epilogue (shared with the fp32 kernel): acc = hi + lo / 2^11
    f32x4 acc[4][NB];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][nb][r] = fmaf(lo[m][nb][r], 1.0f / CS_LO, hi[m][nb][r]);
    conv_box_epilogue<8, 8, 8, 1, 8, 4, NB, (size_t)CS_LDS_BYTES>(a, acc, reinterpret_cast<float*>(lds), tid, lane, wave, n0, z0, y0, x0, cob, lblock);
}

// ---------------------------------------------------------------------------------------------------------- 4^3 volumes
// The same layer on whole 4^3 samples (the retrieval backbone's third level, n = B*K*64 patches): 8 samples per workgroup, wave w = sample
// w, m-block = one z-plane (16 voxels), so the accumulator tiles are in the plain (sample, z, y, x) order of conv_box.h and its epilogue
// applies.  LDS image: 8 halo cubes of 6^3 16-byte slots (8 channels of a voxel), h and l plane; the ring of every cube is the zero padding
// of the normalised tensor and is written once.  A thread stages ONE voxel per chunk (8 channel loads, requested before the previous
// chunk's MFMAs), so the staging is a tenth of the 8^3 box kernel's per MFMA.  42 % of the (voxel, tap) pairs of a 4^3 volume read
// padding: the position-major fp32 kernel (conv3d_small.hip) issues none of them, this one issues them as zeros except where a whole
// (z-plane, k-step) is padding -- at 3 x 16 cycles per k-step against 8 x 32 that still leaves the F16 pipe 2.5x ahead.
namespace {
constexpr int S4_SLOTS = 8 * 216;
constexpr int S4_PLANE = S4_SLOTS * 16;                          // 27,648 bytes
constexpr int S4_LDS_BYTES = 2 * S4_PLANE;                       // 55,296: one chunk image; two workgroups per CU
}   // namespace

// NB n-blocks per workgroup: 1 (four waves per SIMD; every 16-cout block of a layer stages and converts the samples again) or all of a 32 / 64-cout
// layer's (two waves per SIMD, up to 256 VGPRs: staged once)
template <int NB>
__global__ __launch_bounds__(512, NB == 1 ? 4 : 2) void k_conv3_split_s4(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cin = a.c0, nC = cin >> 3;
    const unsigned lblock = rf_xcd_contiguous(blockIdx.x, gridDim.x);
    const int n0 = (int)lblock * 8;
    const int cob = blockIdx.y * (NB * 16);
    const int ns = n0 + wave < a.n ? n0 + wave : a.n - 1;         // ragged last group: re-reads the last sample, the epilogue masks its stores
    const float4* __restrict__ aff = a.affine + (size_t)ns * cin;
    const float* __restrict__ s0 = a.src0 + (size_t)ns * cin * 64 + lane;             // this thread's voxel (z, y, x) = (lane >> 4, (lane >> 2) & 3, lane & 3)
    // halo cube [6][6][6] in 16-byte slots, Y-MAJOR: slot(z, y, x) = y * 36 + z * 6 + x.  An A operand is a ds_read_b128 of an m-block = one z plane (4 y x 4 x);
    // with the rows of the plane 36 = 4 (mod 16) slots apart its 16 voxels sit on 16 different slot positions (z-major, rows 6 apart: rows 0 and 3 collide --
    // tools/lds_bank_model.py's count: 2.07 -> 1.71 LDS cycles per 16-lane group over the 7 k-steps; what is left is where two taps meet in one group)
    unsigned char* const myslot = lds + (wave * 216 + ((lane >> 4) + 1) * 6 + (((lane >> 2) & 3) + 1) * 36 + (lane & 3) + 1) * 16;

    auto stage_load = [&](float (&x)[8], int ca) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = s0[(size_t)(ca * 8 + j) * 64];
    };
    auto stage_store = [&](const float (&x)[8], int ca) {
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 af = aff[ca * 8 + j];                      // wave-uniform: scalar loads
            y[j] = fmaf(x[j] - af.x, af.y, af.z);
        }
        h8 h, l;
        cs_split8(y, h, l);
        *reinterpret_cast<h8*>(myslot) = h;
        *reinterpret_cast<h8*>(myslot + S4_PLANE) = l;
    };

    float xr[8];
    stage_load(xr, 0);
    for (int i = tid; i < S4_LDS_BYTES / 16; i += 512) *reinterpret_cast<float4*>(lds + i * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

    // operand addressing: row i of m-block m = voxel (z = m, y = i >> 2, x = i & 3) of sample `wave`; lane group g supplies tap 4 s + g
    const int g = lane >> 4, ri = lane & 15;
    const unsigned char* const abase = lds + (wave * 216 + 6 + ((ri >> 2) + 1) * 36 + (ri & 3) + 1) * 16;
    int atap[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int tp = 4 * s + g < 27 ? 4 * s + g : 26;             // tap 27: zero weights
        atap[s] = ((tp / 9 - 1) * 6 + ((tp / 3) % 3 - 1) * 36 + (tp % 3 - 1)) * 16;
    }
    f32x4 hi[4][NB], lo[4][NB];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { hi[m][nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; lo[m][nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int nbt = a.cout16 >> 4;
    const h8* __restrict__ wn = reinterpret_cast<const h8*>(a.wp) + (size_t)blockIdx.y * NB * 128 + lane;       // next k-step to fetch
    const int wstep = nbt * 128;
    h8 bh[NB], bl[NB], nh[NB], nl[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { bh[nb] = wn[nb * 128]; bl[nb] = wn[nb * 128 + 64]; }
    wn += wstep;
    __syncthreads();                                                // the zero fill is complete
    stage_store(xr, 0);
    __syncthreads();

    for (int ca = 0; ca < nC; ++ca) {
        const bool more = ca + 1 < nC;
        if (more) stage_load(xr, ca + 1);                           // lands under this chunk's MFMAs
#pragma unroll
        for (int s = 0; s < 7; ++s) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) { nh[nb] = wn[nb * 128]; nl[nb] = wn[nb * 128 + 64]; }      // next k-step's weights (the image has one k-step of slack)
            wn += wstep;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if ((m == 0 && s < 2) || (m == 3 && s >= 5)) continue;      // every tap of the k-step reads the plane below / above the volume: zeros
                const h8 ah = *reinterpret_cast<const h8*>(abase + m * 96 + atap[s]);
                const h8 al = *reinterpret_cast<const h8*>(abase + m * 96 + atap[s] + S4_PLANE);
                cs_mfma_block<NB>(hi[m], lo[m], ah, al, bh, bl);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) { bh[nb] = nh[nb]; bl[nb] = nl[nb]; }
        }
        __syncthreads();                                            // everyone left the image
        if (more) {
            stage_store(xr, ca + 1);
            __syncthreads();
        }
    }

    f32x4 acc[4][NB];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][nb][r] = fmaf(lo[m][nb][r], 1.0f / CS_LO, hi[m][nb][r]);
    conv_box_epilogue<4, 4, 4, 8, 8, 4, NB, (size_t)S4_LDS_BYTES>(a, acc, reinterpret_cast<float*>(lds), tid, lane, wave, n0, 0, 0, 0, cob, lblock);
}

// ------------------------------------------------------------------------------------------- pre-split activations: a producer
// Pre-split tensor of C channels on S^3 voxels: [n][cg = C / 8][plane h | plane l][voxel][8 halves] -- per (sample, 8-channel group) the LDS
// image of the split box kernel in global memory: as many bytes as the fp32 tensor, but already normalised by the CONSUMER's GroupNorm,
// scaled and split, so the consumer stages it with copies.  A producer can write it when one workgroup sees a whole sample (it needs the
// sample's statistics before it can normalise).
//
// k_conv3_cin1_presplit: the first conv of a U-Net's level-0 DoubleConv (1 -> 8 channels, model/unet.py:125-144) on whole 16^3 samples, one
// workgroup of 512 threads per sample; thread = one (y, x) column of 8 z voxels, all 8 couts in registers (cout pairs on v_pk_fma_f32, taps
// accumulated in the (dy, dx, dz) order of k_conv3_cin1).  The input's own GroupNorm (one channel) is computed here too: the workgroup holds the
// whole sample.  Then: ReLU, per-channel
// sums in float64 (recursive halving over the lanes, fixed order), the NEXT layer's GroupNorm triple (gn_affine), normalise, split, store --
// a thread owns whole slots (8 channels of a voxel), and the 256 threads of a z plane write 4 KB contiguously.
struct Cin1PreArgs {
    const float* src;          // [n][16^3]
    const float* gamma_in;     // GroupNorm of the input (cin = 1: one channel, one group): weight [1], bias [1]
    const float* beta_in;
    double eps_in;
    const float* wp;           // conv3 weight image [27][cin4][cout16]
    int n, cin4, cout16;
    const float* gamma;        // the consumer's GroupNorm over the 8 output channels
    const float* beta;
    int cpg;                   // channels per group there (1, 2, 4 or 8)
    double eps;
    unsigned char* out;        // pre-split [n][1][2][4096][8 halves]
};

__global__ __launch_bounds__(512, 4) void k_conv3_cin1_presplit(Cin1PreArgs a) {
    constexpr int E = 16, H = 18, VOL = E * E * E;
    __shared__ float xs[H * H * H];
    __shared__ __attribute__((aligned(16))) float wl[27 * 8];
    __shared__ double red[8 * 16];
    __shared__ float4 nxt[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nn = blockIdx.x;
    for (int i = tid; i < H * H * H; i += 512) xs[i] = 0.f;
    for (int i = tid; i < 27 * 8; i += 512) wl[i] = a.wp[(size_t)(i / 8) * a.cin4 * a.cout16 + (i % 8)];
    const float* src = a.src + (size_t)nn * VOL;
    float raw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) raw[i] = src[tid + i * 512];
    {   // GroupNorm of the input sample (1 channel): float64 sums, wave butterflies, the 8 waves in order
        double sm = 0.0, sq = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { sm += (double)raw[i]; sq += (double)raw[i] * raw[i]; }
        sm = wave_sum(sm); sq = wave_sum(sq);
        if (lane == 0) { red[2 * wave] = sm; red[2 * wave + 1] = sq; }
    }
    __syncthreads();
    float4 af;
    {
        double sm = 0.0, sq = 0.0;
        for (int w = 0; w < 8; ++w) { sm += red[2 * w]; sq += red[2 * w + 1]; }
        const double mean = sm / VOL;
        double var = sq / VOL - mean * mean;
        if (var < 0.0) var = 0.0;
        af = gn_affine(mean, 1.0 / sqrt(var + a.eps_in), a.gamma_in[0], a.beta_in[0]);
    }
    __syncthreads();                                                // red is reused for the output statistics
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int v = tid + i * 512;
        xs[((v >> 8) + 1) * (H * H) + (((v >> 4) & 15) + 1) * H + (v & 15) + 1] = fmaf(raw[i] - af.x, af.y, af.z);
    }
    __syncthreads();
    const int x = tid & 15, y = (tid >> 4) & 15, z0 = (tid >> 8) * 8;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[8][4];
#pragma unroll
    for (int z = 0; z < 8; ++z)
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) acc2[z][cp] = (f32x2){0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            float col[10];
#pragma unroll
            for (int hz = 0; hz < 10; ++hz) col[hz] = xs[(z0 + hz) * (H * H) + (y + dy) * H + x + dx];
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) {
                const float4 w0 = *reinterpret_cast<const float4*>(wl + ((dz * 3 + dy) * 3 + dx) * 8);
                const float4 w1 = *reinterpret_cast<const float4*>(wl + ((dz * 3 + dy) * 3 + dx) * 8 + 4);
                const f32x2 wv[4] = {(f32x2){w0.x, w0.y}, (f32x2){w0.z, w0.w}, (f32x2){w1.x, w1.y}, (f32x2){w1.z, w1.w}};
#pragma unroll
                for (int z = 0; z < 8; ++z) {
                    const f32x2 c2 = (f32x2){col[z + dz], col[z + dz]};
#pragma unroll
                    for (int cp = 0; cp < 4; ++cp) acc2[z][cp] = __builtin_elementwise_fma(c2, wv[cp], acc2[z][cp]);
                }
            }
        }
    float act[8][8];                                                // ReLU'd outputs [z][cout]
#pragma unroll
    for (int z = 0; z < 8; ++z)
#pragma unroll
        for (int co = 0; co < 8; ++co) act[z][co] = fmaxf(acc2[z][co >> 1][co & 1], 0.f);
    {   // statistics of the sample: per (wave, cout) float64 sums by recursive halving, then the 8 waves in order
        double v[16];
#pragma unroll
        for (int co = 0; co < 8; ++co) {
            double sm = 0.0, sq = 0.0;
#pragma unroll
            for (int z = 0; z < 8; ++z) { const double t = (double)act[z][co]; sm += t; sq += t * t; }
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
    }
    __syncthreads();
    if (tid < 8) {                                                  // the consumer's GroupNorm triple of channel tid
        const int g0 = tid / a.cpg * a.cpg;
        double sm = 0.0, sq = 0.0;
        for (int c = g0; c < g0 + a.cpg; ++c)
            for (int w = 0; w < 8; ++w) { sm += red[w * 16 + 2 * c]; sq += red[w * 16 + 2 * c + 1]; }
        const double count = (double)a.cpg * VOL, mean = sm / count;
        double var = sq / count - mean * mean;
        if (var < 0.0) var = 0.0;
        nxt[tid] = gn_affine(mean, 1.0 / sqrt(var + a.eps), a.gamma[tid], a.beta[tid]);
    }
    __syncthreads();
    // normalise + split on cout PAIRS: the 2^-4 activation scale is folded into the triple (exact: a power of two commutes with the fma's rounding), the
    // GroupNorm apply runs as v_pk_add_f32 / v_pk_fma_f32 on the pairs the accumulators already are, and both conversions are v_cvt_pk_f16_f32 --
    // 3.5 VALU instructions per value instead of 7 (this kernel is VALU-bound: DESIGN 10)
    f32x2 negc[4], sc[4], sh[4];
#pragma unroll
    for (int cp = 0; cp < 4; ++cp) {
        const float4 t0 = nxt[2 * cp], t1 = nxt[2 * cp + 1];
        negc[cp] = (f32x2){-t0.x, -t1.x};
        sc[cp] = (f32x2){t0.y * CS_ACT_SCALE, t1.y * CS_ACT_SCALE};
        sh[cp] = (f32x2){t0.z * CS_ACT_SCALE, t1.z * CS_ACT_SCALE};
    }
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    unsigned char* outp = a.out + (size_t)nn * 2 * VOL * 16;
#pragma unroll
    for (int z = 0; z < 8; ++z) {
        h8 h, l;
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) {
            const f32x2 relu = (f32x2){act[z][2 * cp], act[z][2 * cp + 1]};
            f32x2 v = __builtin_elementwise_fma(relu + negc[cp], sc[cp], sh[cp]);
            v[0] = __builtin_amdgcn_fmed3f(v[0], -65504.f, 65504.f);
            v[1] = __builtin_amdgcn_fmed3f(v[1], -65504.f, 65504.f);
            const h2 hh = __builtin_convertvector(v, h2);
            const f32x2 back = __builtin_convertvector(hh, f32x2);
            const h2 ll = __builtin_convertvector((v - back) * CS_LO, h2);      // v - h is exact in fp32
            h[2 * cp] = hh[0]; h[2 * cp + 1] = hh[1];
            l[2 * cp] = ll[0]; l[2 * cp + 1] = ll[1];
        }
        const size_t vox = ((size_t)(z0 + z) * E + y) * E + x;
        *reinterpret_cast<h8*>(outp + vox * 16) = h;
        *reinterpret_cast<h8*>(outp + ((size_t)VOL + vox) * 16) = l;
    }
}

extern "C" size_t rf_split_act_bytes(int n, int c, int edge) {
    return (size_t)n * (size_t)((c + 7) / 8) * 2 * (size_t)edge * edge * edge * 16;
}

extern "C" int rf_conv3d_cin1_presplit_supported(int n, int edge, int cout, int next_groups) {
    return n > 0 && edge == 16 && cout == 8 && next_groups > 0 && 8 % next_groups == 0;
}

extern "C" int rf_conv3d_cin1_presplit(const float* src, int n, int edge, const float* in_gamma, const float* in_beta, float in_eps, const float* w_packed,
                                       int cout, const float* next_gamma, const float* next_beta, int next_groups, float eps, void* out_presplit,
                                       void* stream) {
    RF_REQUIRE(rf_conv3d_cin1_presplit_supported(n, edge, cout, next_groups), RF_E_UNSUPPORTED,
               "rf_conv3d_cin1_presplit: takes 1 -> 8 channels on 16^3 samples with 1, 2, 4 or 8 groups in the consumer's GroupNorm (got n=%d edge=%d cout=%d groups=%d)",
               n, edge, cout, next_groups);
    RF_REQUIRE(src && in_gamma && in_beta && w_packed && next_gamma && next_beta && out_presplit, RF_E_INVALID, "rf_conv3d_cin1_presplit: null pointer");
    Cin1PreArgs a;
    a.src = src; a.gamma_in = in_gamma; a.beta_in = in_beta; a.eps_in = (double)in_eps; a.wp = w_packed; a.n = n; a.cin4 = 4; a.cout16 = 16;
    a.gamma = next_gamma; a.beta = next_beta; a.cpg = 8 / next_groups; a.eps = (double)eps; a.out = reinterpret_cast<unsigned char*>(out_presplit);
    hipLaunchKernelGGL(k_conv3_cin1_presplit, dim3((unsigned)n), dim3(512), 0, (hipStream_t)stream, a);
    RF_CHECK_LAUNCH("rf_conv3d_cin1_presplit");
    return RF_OK;
}

// -------------------------------------------------------------------------------------------------------------------- host
extern "C" int rf_conv3d_split_supported(int c0, int c1, int n, int edge, int cout) {
    // whole 4^3 samples (8 per workgroup, k_conv3_split_s4): cin in eights, any cout (16 per workgroup), enough samples to fill the chip
    if (edge == 4) return c1 == 0 && c0 >= 8 && c0 % 8 == 0 && n >= 1024 && cout > 0;
    // channel counts that are not multiples of 8 are padded up with zero channels: taken when at least 3/4 of the slots are real (6, 12, 20, 28, 42 ...)
    if (c1 != 0 || c0 < 6 || 4 * c0 < 3 * rf_round_up(c0, 8) || n <= 0 || cout <= 0 || !rf_is_pow2(edge) || edge < 8 || edge > 128) return 0;
    const int cout16 = rf_round_up(cout, 16);
    // >= 256 boxes (rf_conv_use_big asks for 1024 workgroups: a chip-filling launch): the chunk-level U-Net's 32 -> 32 @16^3 conv at B = 32 is 256 boxes and took
    // 0.25 ms of a 1024-workgroup fp32-MFMA launch beside the retrieval path; the box kernel does it in a fraction of that on the F16 cores
    const long long boxes = (long long)n * (edge / 8) * (edge / 8) * (edge / 8);
    if (cout16 <= 32) return rf_conv_use_big(n, edge, cout16) || boxes >= 256;
    // more couts (round 6; the deep levels of the chunk-level U-Nets: 24 -> 48 @32^3, 48 -> 96 and 96 -> 96 @16^3 at 16 chunks): cout blocks of 16 / 32 on grid.y, every block
    // staging the box again -- still a third of the fp32-MFMA form's time (C5: 0.33 / 0.18 / 0.34 ms) as long as the launch has a couple of workgroups per CU
#ifdef RF_SPLIT_NARROW_ONLY                                          // dev A/B (tools/build_variant.py): rounds 2-5's rule
    return 0;
#endif
    return cout16 <= 192 && boxes * (cout16 / 16) >= 512;
}

template <int NB, int WPS, bool ONE, bool PADC = false, bool PRE = false>
static int launch_split(const ConvArgs& a, hipStream_t stream, const SplitPreOut& po = SplitPreOut{nullptr, nullptr, nullptr, 0, 0.f, nullptr, nullptr, nullptr, 0.f, 0.f}) {
    auto kern = k_conv3_split<NB, WPS, ONE, PADC, PRE>;
    if constexpr (!ONE) {                                             // two chunk buffers: 66,560 bytes, past the 64 KB a kernel gets without asking
        static RfLdsOptIn opt;
        if (int rc = opt.ensure(reinterpret_cast<const void*>(kern), CS_LDS_BYTES, "rf_conv3d_split_k3_gn_relu")) return rc;
    }
    const unsigned gx = (unsigned)a.n * (a.edge / 8) * (a.edge / 8) * (a.edge / 8);
    hipLaunchKernelGGL(kern, dim3(gx, (unsigned)(a.cout16 / (NB * 16))), dim3(512), ONE ? CS_BUF : CS_LDS_BYTES, stream, a, po);
    RF_CHECK_LAUNCH("rf_conv3d_split_k3_gn_relu");
    return RF_OK;
}

static int split_run(const ConvArgs& a, void* stream) {
    // 32 couts: one workgroup with two n-blocks per box (158 VGPRs: every A operand read from LDS feeds six MFMAs instead of three -- the
    // 16-cout instance asks the LDS for 170 B/clk of A operands and gets 128).  Round 2 ran 32 couts as two 16-cout workgroups because this
    // instance leaves room for other kernels' waves on its SIMDs and fp32 kernels of another stream then returned different bits beside its
    // F16 MFMAs; that was an unsafe packed-fp32 instruction form in THOSE kernels, gone since round 3.
    // one 8-channel chunk (the retrieval backbone's 8 -> 16 @16^3 conv): no prefetch registers, one chunk buffer (32 KB), 80 VGPRs -> three
    // workgroups = six waves per SIMD per CU; the layer is bound by the per-box latency chain (load -> stage -> 7 k-steps -> epilogue), and
    // a third box in flight per CU is worth 13 % (1.60 -> 1.40 ms)
    const int cin = a.c0, n = a.n, edge = a.edge;
    if (edge == 4) {
        RF_REQUIRE(!a.pool_out, RF_E_UNSUPPORTED, "rf_conv3d_split_k3_gn_relu: the 4^3 form has no fused max-pool (pool its output with rf_maxpool3d_2_stats)");
        // all couts of a 32 / 64-cout layer in one workgroup (samples staged once; 4^3 levels of the retrieval backbone: 32 -> 32, 32 -> 64, 64 -> 64)
        const unsigned gx = (unsigned)((n + 7) / 8), nbt = (unsigned)(a.cout16 / 16);
        if (CS_S4_WIDE && nbt % 4 == 0) hipLaunchKernelGGL(k_conv3_split_s4<4>, dim3(gx, nbt / 4), dim3(512), S4_LDS_BYTES, (hipStream_t)stream, a);
        else if (CS_S4_WIDE && nbt % 3 == 0) hipLaunchKernelGGL(k_conv3_split_s4<3>, dim3(gx, nbt / 3), dim3(512), S4_LDS_BYTES, (hipStream_t)stream, a);      // nf = 12: 48 / 96 couts
        else if (CS_S4_WIDE && nbt % 2 == 0) hipLaunchKernelGGL(k_conv3_split_s4<2>, dim3(gx, nbt / 2), dim3(512), S4_LDS_BYTES, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(k_conv3_split_s4<1>, dim3(gx, nbt), dim3(512), S4_LDS_BYTES, (hipStream_t)stream, a);
        RF_CHECK_LAUNCH("rf_conv3d_split_k3_gn_relu");
        return RF_OK;
    }
    if (a.cout16 <= 32 && rf_split_zcm_takes(cin, n, edge, a.cout))
        return rf_split_zcm_launch(a, SplitPreOut{nullptr, nullptr, nullptr, 0, 0.f, nullptr, nullptr, nullptr, 0.f, 0.f}, false, (hipStream_t)stream, "rf_conv3d_split_k3_gn_relu");
    if (cin % 8) {
        if (cin < 8) return launch_split<1, 6, true, true>(a, (hipStream_t)stream);                   // 6 -> 12 of the nf = 12 U-Nets: one chunk, two zero slots
        if (a.cout16 % 32 == 0) return launch_split<2, 2, false, true>(a, (hipStream_t)stream);
        return launch_split<1, 4, false, true>(a, (hipStream_t)stream);
    }
    if (cin > 8 && a.cout16 % 32 == 0) return launch_split<2, 2, false>(a, (hipStream_t)stream);
    return cin == 8 ? launch_split<1, 6, true>(a, (hipStream_t)stream) : launch_split<1, 4, false>(a, (hipStream_t)stream);
}

extern "C" int rf_conv3d_split_k3_gn_relu(const float* src, int cin, int n, int edge, const float* gn_affine, const void* w_packed, int cout,
                                           float* out, double* stats, float* pool_out, double* pool_stats, void* stream) {
    RF_REQUIRE(rf_conv3d_split_supported(cin, 0, n, edge, cout), RF_E_UNSUPPORTED,
               "rf_conv3d_split_k3_gn_relu: takes cin >= 6 (at least 3/4 of the next multiple of 8), up to 192 couts, edge >= 8 and enough 8^3 boxes (got cin=%d n=%d edge=%d cout=%d)",
               cin, n, edge, cout);
    RF_REQUIRE(src && gn_affine && w_packed && (out || pool_out), RF_E_INVALID, "rf_conv3d_split_k3_gn_relu: null pointer");
    RF_REQUIRE(out || !stats, RF_E_INVALID, "rf_conv3d_split_k3_gn_relu: statistics of an output that is not written");
    RF_REQUIRE(pool_out || !pool_stats, RF_E_INVALID, "rf_conv3d_split_k3_gn_relu: pooled statistics without a pooled output");
    ConvArgs a;
    a.src0 = src; a.src1 = nullptr; a.affine = reinterpret_cast<const float4*>(gn_affine); a.wp = reinterpret_cast<const float*>(w_packed); a.out = out;
    a.c0 = cin; a.c1 = 0; a.n = n; a.edge = edge; a.cout = cout; a.cin4 = cin; a.cout16 = rf_round_up(cout, 16);
    a.stats = reinterpret_cast<double2*>(stats);
    a.stats_tiles = (stats || pool_stats) ? (edge == 4 ? 1 : (edge / 8) * (edge / 8) * (edge / 8)) : 0;
    a.pool_out = pool_out; a.pool_stats = reinterpret_cast<double2*>(pool_stats);
    a.pool_mode = pool_out ? (out ? 1 : 2) : 0;
    a.floor = 0.f;
    return split_run(a, stream);
}

// The same conv as a plain operator for the training slice's data gradient (rfuse/autograd.py: d xn = conv3(dz, W^T with flipped taps)): any cout
// (16 or 32 per workgroup, the cout blocks on grid.y), ReLU optional, no statistics, no pooling.  The caller scales dz into the split forms' range
// through the affine (rf_dgrad_scale_affine: a power of two, exact) and takes the scale out again downstream.
extern "C" int rf_conv3d_split_k3_gn_supported(int cin, int n, int edge, int cout) {
    if (edge == 4) return rf_conv3d_split_supported(cin, 0, n, edge, cout);
    if (cin < 6 || 4 * cin < 3 * rf_round_up(cin, 8) || n <= 0 || cout <= 0 || !rf_is_pow2(edge) || edge < 8 || edge > 128) return 0;
    return rf_conv_use_big(n, edge, 64);
}

extern "C" int rf_conv3d_split_k3_gn(const float* src, int cin, int n, int edge, const float* gn_affine, const void* w_packed, int cout, int relu,
                                      float* out, void* stream) {
    RF_REQUIRE(rf_conv3d_split_k3_gn_supported(cin, n, edge, cout), RF_E_UNSUPPORTED,
               "rf_conv3d_split_k3_gn: takes cin >= 6 (at least 3/4 of the next multiple of 8), edge >= 8 with enough 8^3 boxes or whole 4^3 samples (got cin=%d n=%d edge=%d cout=%d)",
               cin, n, edge, cout);
    RF_REQUIRE(src && gn_affine && w_packed && out, RF_E_INVALID, "rf_conv3d_split_k3_gn: null pointer");
    ConvArgs a;
    a.src0 = src; a.src1 = nullptr; a.affine = reinterpret_cast<const float4*>(gn_affine); a.wp = reinterpret_cast<const float*>(w_packed); a.out = out;
    a.c0 = cin; a.c1 = 0; a.n = n; a.edge = edge; a.cout = cout; a.cin4 = cin; a.cout16 = rf_round_up(cout, 16);
    a.stats = nullptr; a.stats_tiles = 0; a.pool_out = nullptr; a.pool_stats = nullptr; a.pool_mode = 0;
    a.floor = relu ? 0.f : -INFINITY;
    return split_run(a, stream);
}

// The same layer on a PRE-SPLIT input (already normalised for this layer's GroupNorm and split by its producer: rf_conv3d_cin1_presplit):
// cin in whole 8-channel groups, 16 couts per workgroup (32: two n-blocks), edge >= 8.
extern "C" int rf_conv3d_split_pre_supported(int cin, int n, int edge, int cout) {
    return cin >= 8 && cin % 8 == 0 && edge >= 8 && rf_conv3d_split_supported(cin, 0, n, edge, cout);
}


// tiles per (sample, cout) of the statistics rf_conv3d_split_pre_k3_relu writes for this shape ([n][cout][tiles] double2): one per 8^3 box, or one per
// sample where the persistent form (which sums a sample's boxes itself) takes the layer
extern "C" int rf_conv3d_split_pre_stats_tiles(int cin, int n, int edge, int cout) {
    if (rf_split_zc_takes(cin, n, edge, cout)) return 1;
    return (edge / 8) * (edge / 8) * (edge / 8);
}

extern "C" int rf_conv3d_split_pre_k3_relu(const void* src_presplit, int cin, int n, int edge, const void* w_packed, int cout,
                                           float* out, double* stats, float* pool_out, double* pool_stats, void* stream) {
    RF_REQUIRE(rf_conv3d_split_pre_supported(cin, n, edge, cout), RF_E_UNSUPPORTED,
               "rf_conv3d_split_pre_k3_relu: takes cin in eights, up to 32 couts, edge >= 8 and enough 8^3 boxes (got cin=%d n=%d edge=%d cout=%d)", cin, n, edge, cout);
    RF_REQUIRE(src_presplit && w_packed && (out || pool_out), RF_E_INVALID, "rf_conv3d_split_pre_k3_relu: null pointer");
    RF_REQUIRE(out || !stats, RF_E_INVALID, "rf_conv3d_split_pre_k3_relu: statistics of an output that is not written");
    RF_REQUIRE(pool_out || !pool_stats, RF_E_INVALID, "rf_conv3d_split_pre_k3_relu: pooled statistics without a pooled output");
    ConvArgs a;
    a.src0 = reinterpret_cast<const float*>(src_presplit); a.src1 = nullptr; a.affine = nullptr; a.wp = reinterpret_cast<const float*>(w_packed); a.out = out;
    a.c0 = cin; a.c1 = 0; a.n = n; a.edge = edge; a.cout = cout; a.cin4 = cin; a.cout16 = rf_round_up(cout, 16);
    a.stats = reinterpret_cast<double2*>(stats);
    a.stats_tiles = (stats || pool_stats) ? (edge / 8) * (edge / 8) * (edge / 8) : 0;
    a.pool_out = pool_out; a.pool_stats = reinterpret_cast<double2*>(pool_stats);
    a.pool_mode = pool_out ? (out ? 1 : 2) : 0;
    a.floor = 0.f;
    if (rf_split_zc_takes(cin, n, edge, cout)) {
        a.stats_tiles = 1;
        return rf_split_zc_launch(a, SplitPreOut{nullptr, nullptr, nullptr, 0, 0.f, nullptr, nullptr, nullptr, 0.f, 0.f}, (hipStream_t)stream);
    }
    if (a.cout16 <= 32 && rf_split_zcm_takes(cin, n, edge, cout))
        return rf_split_zcm_launch(a, SplitPreOut{nullptr, nullptr, nullptr, 0, 0.f, nullptr, nullptr, nullptr, 0.f, 0.f}, true, (hipStream_t)stream, "rf_conv3d_split_pre_k3_relu");
    if (cin > 8 && a.cout16 == 32) return launch_split<2, 2, false, false, true>(a, (hipStream_t)stream);
    return cin == 8 ? launch_split<1, 6, true, false, true>(a, (hipStream_t)stream) : launch_split<1, 4, false, false, true>(a, (hipStream_t)stream);
}

// rf_conv3d_split_pre_k3_relu on a pre-split input of whole 8^3 samples whose voxel slots are in PARITY-MAJOR order (rf_conv3d_up_split_presplit_pm): the
// persistent multi-chunk form only (>= 2048 samples, cin >= 16 in eights, <= 32 couts)
extern "C" int rf_conv3d_split_pre_pm_supported(int cin, int n, int edge, int cout) {
    return edge == 8 && rf_conv3d_split_pre_supported(cin, n, edge, cout) && !rf_split_zc_takes(cin, n, edge, cout) && rf_round_up(cout, 16) <= 32 && rf_split_zcm_takes(cin, n, edge, cout);
}

extern "C" int rf_conv3d_split_pre_pm_k3_relu(const void* src_presplit_pm, int cin, int n, int edge, const void* w_packed, int cout,
                                              float* out, double* stats, float* pool_out, double* pool_stats, void* stream) {
    RF_REQUIRE(rf_conv3d_split_pre_pm_supported(cin, n, edge, cout), RF_E_UNSUPPORTED,
               "rf_conv3d_split_pre_pm_k3_relu: takes whole 8^3 samples (n >= 2048), cin >= 16 in eights, up to 32 couts (got cin=%d n=%d edge=%d cout=%d)", cin, n, edge, cout);
    RF_REQUIRE(src_presplit_pm && w_packed && (out || pool_out), RF_E_INVALID, "rf_conv3d_split_pre_pm_k3_relu: null pointer");
    RF_REQUIRE(out || !stats, RF_E_INVALID, "rf_conv3d_split_pre_pm_k3_relu: statistics of an output that is not written");
    RF_REQUIRE(pool_out || !pool_stats, RF_E_INVALID, "rf_conv3d_split_pre_pm_k3_relu: pooled statistics without a pooled output");
    ConvArgs a;
    a.src0 = reinterpret_cast<const float*>(src_presplit_pm); a.src1 = nullptr; a.affine = nullptr; a.wp = reinterpret_cast<const float*>(w_packed); a.out = out;
    a.c0 = cin; a.c1 = 0; a.n = n; a.edge = edge; a.cout = cout; a.cin4 = cin; a.cout16 = rf_round_up(cout, 16);
    a.stats = reinterpret_cast<double2*>(stats);
    a.stats_tiles = (stats || pool_stats) ? 1 : 0;
    a.pool_out = pool_out; a.pool_stats = reinterpret_cast<double2*>(pool_stats);
    a.pool_mode = pool_out ? (out ? 1 : 2) : 0;
    a.floor = 0.f;
    a.src_pm = 1;
    return rf_split_zcm_launch(a, SplitPreOut{nullptr, nullptr, nullptr, 0, 0.f, nullptr, nullptr, nullptr, 0.f, 0.f}, true, (hipStream_t)stream, "rf_conv3d_split_pre_pm_k3_relu");
}

// rf_conv3d_split_k3_gn_relu on whole 8^3 samples with up to 16 couts, its output handed to the NEXT SingleConv pre-split: the workgroup holds
// the sample, so it applies the next layer's GroupNorm (next_gamma / next_beta [cout], next_groups, eps) from the sample's own statistics, splits and
// writes rf_split_act_bytes(n, cout, 8) bytes for rf_conv3d_split_pre_k3_relu; the fp32 output is not written.  stats (optional): [n][cout] (sum, sum of squares).
extern "C" int rf_conv3d_split_presplit_supported(int cin, int n, int edge, int cout, int next_groups) {
    return edge == 8 && cin >= 16 && cin % 8 == 0 && (cout == 8 || cout == 16) && next_groups > 0 && cout % next_groups == 0 && rf_conv3d_split_supported(cin, 0, n, edge, cout);
}

extern "C" int rf_conv3d_split_presplit(const float* src, int cin, int n, int edge, const float* gn_affine, const void* w_packed, int cout, const float* next_gamma,
                                        const float* next_beta, int next_groups, float eps, void* out_presplit, double* stats, void* stream) {
    RF_REQUIRE(rf_conv3d_split_presplit_supported(cin, n, edge, cout, next_groups), RF_E_UNSUPPORTED,
               "rf_conv3d_split_presplit: takes whole 8^3 samples, cin >= 16 in eights, 8 or 16 couts in whole groups (got cin=%d n=%d edge=%d cout=%d groups=%d)", cin, n, edge, cout, next_groups);
    RF_REQUIRE(src && gn_affine && w_packed && next_gamma && next_beta && out_presplit, RF_E_INVALID, "rf_conv3d_split_presplit: null pointer");
    ConvArgs a;
    a.src0 = src; a.src1 = nullptr; a.affine = reinterpret_cast<const float4*>(gn_affine); a.wp = reinterpret_cast<const float*>(w_packed); a.out = nullptr;
    a.c0 = cin; a.c1 = 0; a.n = n; a.edge = edge; a.cout = cout; a.cin4 = cin; a.cout16 = 16;
    a.stats = reinterpret_cast<double2*>(stats); a.stats_tiles = stats ? 1 : 0;
    a.pool_out = nullptr; a.pool_stats = nullptr; a.pool_mode = 0; a.floor = 0.f;
    const SplitPreOut po{reinterpret_cast<h8*>(out_presplit), next_gamma, next_beta, next_groups, eps, nullptr, nullptr, nullptr, 0.f, 0.f};
    if (cout <= 16 && rf_split_zcm_takes(cin, n, edge, cout)) return rf_split_zcm_launch(a, po, false, (hipStream_t)stream, "rf_conv3d_split_presplit");
    return launch_split<1, 4, false>(a, (hipStream_t)stream, po);
}

// rf_conv3d_split_k3_gn_relu with the final decoder's head fused into the epilogue (reference model/refinement.py:48-61 + trainer/train_refinement.py:242-243):
// out1 [n][1][edge^3] = (tanh(sum_c pw_w[c] * relu(conv(GN(x)))[c] + pw_b[0]) + post_add) * post_mul -- the arithmetic of rf_conv1x1_tanh on the conv's
// output, bit for bit -- without writing or re-reading the cout-channel tensor.  cin >= 12 (two or more chunks), up to 16 couts.
extern "C" int rf_conv3d_split_pointwise_supported(int cin, int n, int edge, int cout) {
    return cin >= 12 && cout <= 16 && edge >= 8 && rf_conv3d_split_supported(cin, 0, n, edge, cout);
}

extern "C" int rf_conv3d_split_k3_gn_relu_pointwise_tanh(const float* src, int cin, int n, int edge, const float* gn_affine, const void* w_packed, int cout,
                                                         const float* pw_w, const float* pw_b, float post_add, float post_mul, float* out1, void* stream) {
    RF_REQUIRE(rf_conv3d_split_pointwise_supported(cin, n, edge, cout), RF_E_UNSUPPORTED,
               "rf_conv3d_split_k3_gn_relu_pointwise_tanh: takes the shapes of rf_conv3d_split_k3_gn_relu with cin >= 12 and up to 16 couts (got cin=%d n=%d edge=%d cout=%d)", cin, n, edge, cout);
    RF_REQUIRE(src && gn_affine && w_packed && pw_w && pw_b && out1, RF_E_INVALID, "rf_conv3d_split_k3_gn_relu_pointwise_tanh: null pointer");
    ConvArgs a;
    a.src0 = src; a.src1 = nullptr; a.affine = reinterpret_cast<const float4*>(gn_affine); a.wp = reinterpret_cast<const float*>(w_packed); a.out = nullptr;
    a.c0 = cin; a.c1 = 0; a.n = n; a.edge = edge; a.cout = cout; a.cin4 = cin; a.cout16 = 16;
    a.stats = nullptr; a.stats_tiles = 0; a.pool_out = nullptr; a.pool_stats = nullptr; a.pool_mode = 0; a.floor = 0.f;
    const SplitPreOut po{nullptr, nullptr, nullptr, 0, 0.f, out1, pw_w, pw_b, post_add, post_mul};
    if (cout <= 16 && rf_split_zcm_takes(cin, n, edge, cout)) return rf_split_zcm_launch(a, po, false, (hipStream_t)stream, "rf_conv3d_split_k3_gn_relu_pointwise_tanh");
    return cin % 8 ? launch_split<1, 4, false, true>(a, (hipStream_t)stream, po) : launch_split<1, 4, false>(a, (hipStream_t)stream, po);
}

// ... on the CHANNEL-INTERLEAVED output of rf_conv3d_up_split_k3_gn_relu_ch8 ([n][cin / 8][edge^3][8] fp32): same values bit for bit, the staging loads are
// two 16-byte loads per voxel instead of eight 4-byte gathers.  The persistent z-column form only.
extern "C" int rf_conv3d_split_pointwise_ch8_supported(int cin, int n, int edge, int cout) {
    return rf_conv3d_split_pointwise_supported(cin, n, edge, cout) && cin % 8 == 0 && cout <= 16 && rf_split_zcm_takes(cin, n, edge, cout);
}

extern "C" int rf_conv3d_split_k3_gn_relu_pointwise_tanh_ch8(const float* src_ch8, int cin, int n, int edge, const float* gn_affine, const void* w_packed, int cout,
                                                             const float* pw_w, const float* pw_b, float post_add, float post_mul, float* out1, void* stream) {
    RF_REQUIRE(rf_conv3d_split_pointwise_ch8_supported(cin, n, edge, cout), RF_E_UNSUPPORTED,
               "rf_conv3d_split_k3_gn_relu_pointwise_tanh_ch8: takes cin in eights (>= 16), up to 16 couts, edge >= 8 and at least 2048 boxes (got cin=%d n=%d edge=%d cout=%d)", cin, n, edge, cout);
    RF_REQUIRE(src_ch8 && gn_affine && w_packed && pw_w && pw_b && out1, RF_E_INVALID, "rf_conv3d_split_k3_gn_relu_pointwise_tanh_ch8: null pointer");
    ConvArgs a;
    a.src0 = src_ch8; a.src1 = nullptr; a.affine = reinterpret_cast<const float4*>(gn_affine); a.wp = reinterpret_cast<const float*>(w_packed); a.out = nullptr;
    a.c0 = cin; a.c1 = 0; a.n = n; a.edge = edge; a.cout = cout; a.cin4 = cin; a.cout16 = 16;
    a.stats = nullptr; a.stats_tiles = 0; a.pool_out = nullptr; a.pool_stats = nullptr; a.pool_mode = 0; a.floor = 0.f;
    const SplitPreOut po{nullptr, nullptr, nullptr, 0, 0.f, out1, pw_w, pw_b, post_add, post_mul};
    return rf_split_zcm_launch_ch8_pointwise(a, po, (hipStream_t)stream, "rf_conv3d_split_k3_gn_relu_pointwise_tanh_ch8");
}

// rf_conv3d_split_pre_k3_relu of a level-0 second conv (pooled output only) that ALSO hands the pooled tensor to the next level's first conv pre-split: the
// persistent form walks whole 16^3 samples, so at a sample's end it has the statistics of the pooled [cout][8^3] tensor, applies that layer's GroupNorm
// (next_gamma / next_beta [cout], next_groups, eps), splits and writes rf_split_act_bytes(n, cout, 8) bytes for rf_conv3d_split_pre_presplit /
// rf_conv3d_split_pre_k3_relu.  pool_scratch: rf_conv3d_split_pre_pool_presplit_scratch_floats(cout) floats of workspace (a slot per persistent workgroup: the
// pooled fp32 values of the sample in flight, read back through L2 once its statistics are known -- NOT a pooled tensor); pool_stats (optional) as
// rf_conv3d_split_pre_k3_relu.
extern "C" int rf_conv3d_split_pre_pool_presplit_supported(int cin, int n, int edge, int cout, int next_groups) {
    return rf_split_zc_takes(cin, n, edge, cout) && cout % 8 == 0 && next_groups > 0 && cout % next_groups == 0;
}

extern "C" size_t rf_conv3d_split_pre_pool_presplit_scratch_floats(int cout) { return (size_t)rf_persistent_wgs() * (size_t)(cout > 0 ? cout : 0) * 512; }      // workgroups x [cout][8^3]

extern "C" int rf_conv3d_split_pre_k3_relu_pool_presplit(const void* src_presplit, int cin, int n, int edge, const void* w_packed, int cout, float* pool_out,
                                                         double* pool_stats, const float* next_gamma, const float* next_beta, int next_groups, float eps,
                                                         void* out_presplit, void* stream) {
    RF_REQUIRE(rf_conv3d_split_pre_pool_presplit_supported(cin, n, edge, cout, next_groups), RF_E_UNSUPPORTED,
               "rf_conv3d_split_pre_k3_relu_pool_presplit: takes 8 -> 8 or 16 channels on 16^3 samples (n >= 512) with the couts in whole groups (got cin=%d n=%d edge=%d cout=%d groups=%d)",
               cin, n, edge, cout, next_groups);
    RF_REQUIRE(src_presplit && w_packed && pool_out && next_gamma && next_beta && out_presplit, RF_E_INVALID, "rf_conv3d_split_pre_k3_relu_pool_presplit: null pointer");
    ConvArgs a;
    a.src0 = reinterpret_cast<const float*>(src_presplit); a.src1 = nullptr; a.affine = nullptr; a.wp = reinterpret_cast<const float*>(w_packed); a.out = nullptr;
    a.c0 = cin; a.c1 = 0; a.n = n; a.edge = edge; a.cout = cout; a.cin4 = cin; a.cout16 = rf_round_up(cout, 16);
    a.stats = nullptr; a.stats_tiles = 1; a.pool_out = pool_out; a.pool_stats = reinterpret_cast<double2*>(pool_stats); a.pool_mode = 2; a.floor = 0.f;
    const SplitPreOut po{reinterpret_cast<h8*>(out_presplit), next_gamma, next_beta, next_groups, eps, nullptr, nullptr, nullptr, 0.f, 0.f};
    return rf_split_zc_launch(a, po, (hipStream_t)stream);
}

// rf_conv3d_split_presplit on a PRE-SPLIT input: whole 8^3 samples in, whole 8^3 samples out, both as f16 pairs normalised for their consumer's GroupNorm
// (the first conv of the retrieval backbone's level 1 between rf_conv3d_split_pre_k3_relu_pool_presplit and rf_conv3d_split_pre_k3_relu).  The persistent
// z-column form only.  stats (optional): [n][cout] (sum, sum of squares) of the ReLU'd output.
extern "C" int rf_conv3d_split_pre_presplit_supported(int cin, int n, int edge, int cout, int next_groups) {
    return edge == 8 && cin >= 16 && cin % 8 == 0 && (cout == 8 || cout == 16) && next_groups > 0 && cout % next_groups == 0 && rf_split_zcm_takes(cin, n, edge, cout);
}

extern "C" int rf_conv3d_split_pre_presplit(const void* src_presplit, int cin, int n, int edge, const void* w_packed, int cout, const float* next_gamma,
                                            const float* next_beta, int next_groups, float eps, void* out_presplit, double* stats, void* stream) {
    RF_REQUIRE(rf_conv3d_split_pre_presplit_supported(cin, n, edge, cout, next_groups), RF_E_UNSUPPORTED,
               "rf_conv3d_split_pre_presplit: takes whole 8^3 samples (n >= 2048), cin >= 16 in eights, 8 or 16 couts in whole groups (got cin=%d n=%d edge=%d cout=%d groups=%d)", cin, n, edge, cout, next_groups);
    RF_REQUIRE(src_presplit && w_packed && next_gamma && next_beta && out_presplit, RF_E_INVALID, "rf_conv3d_split_pre_presplit: null pointer");
    ConvArgs a;
    a.src0 = reinterpret_cast<const float*>(src_presplit); a.src1 = nullptr; a.affine = nullptr; a.wp = reinterpret_cast<const float*>(w_packed); a.out = nullptr;
    a.c0 = cin; a.c1 = 0; a.n = n; a.edge = edge; a.cout = cout; a.cin4 = cin; a.cout16 = 16;
    a.stats = reinterpret_cast<double2*>(stats); a.stats_tiles = stats ? 1 : 0;
    a.pool_out = nullptr; a.pool_stats = nullptr; a.pool_mode = 0; a.floor = 0.f;
    const SplitPreOut po{reinterpret_cast<h8*>(out_presplit), next_gamma, next_beta, next_groups, eps, nullptr, nullptr, nullptr, 0.f, 0.f};
    return rf_split_zcm_launch(a, po, true, (hipStream_t)stream, "rf_conv3d_split_pre_presplit");
}
