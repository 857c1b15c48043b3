// rf_conv3d_valid_leaky on the matrix cores: valid (no padding) strided Conv3d + bias + LeakyReLU as an fp32-MFMA
// implicit GEMM for gfx950.  Layers of the conv patch encoders: Patch08 (model/retrieval.py:140-147), PCPatch48
// (:221-234, 129 GFLOP per chunk of 64 windows -- more than the refinement network itself), Patch32 / Patch24V2 on the
// database side (:8-19, :339-352); kernel sizes 2..5, strides 1..2, arbitrary (non power-of-two) edges.
//
// GEMM view: M = output voxels of one window (linear index), N = cout, K = k^3 * cin with K index = tap*cin + ci.
// There is no halo tile and no barrier: the windows are small (<= 48^3 x 1..96 channels), every input value is re-read
// k^3 * cout/16 times from L1/L2, so each wave gathers its A operands straight from global memory (16 consecutive
// output voxels -> 16 addresses `stride` floats apart: one or two cache lines) and its B operands from the packed
// weight image [K][cout16]; waves are fully independent and latency is hidden by occupancy (8 waves/SIMD).
// v_mfma_f32_16x16x4_f32 is a k-ordered fp32 FMA chain, so results are plain fp32.
#include "common.h"

__global__ void k_convv_pack(const float* __restrict__ w, int cout, int cin, int k3, int kpad, int cout16, float* __restrict__ wp) {
    const size_t total = (size_t)kpad * cout16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % cout16), kidx = (int)(i / cout16);
        float v = 0.f;
        if (co < cout && kidx < k3 * cin) {
            const int tap = kidx / cin, ci = kidx % cin;
            v = w[((size_t)co * cin + ci) * k3 + tap];
        }
        wp[i] = v;
    }
}

extern "C" size_t rf_convv_packed_floats(int cout, int cin, int k) {
    return (size_t)rf_round_up(k * k * k * cin, 4) * rf_round_up(cout, 16);
}

extern "C" int rf_convv_pack_weight(const float* w_oidhw, int cout, int cin, int k, float* w_packed, void* stream) {
    RF_REQUIRE(w_oidhw && w_packed && cout > 0 && cin > 0 && k > 0, RF_E_INVALID, "rf_convv_pack_weight: bad arguments");
    const size_t total = rf_convv_packed_floats(cout, cin, k);
    const size_t want = (total + 255) / 256;
    hipLaunchKernelGGL(k_convv_pack, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, (hipStream_t)stream, w_oidhw, cout, cin, k * k * k,
                       rf_round_up(k * k * k * cin, 4), rf_round_up(cout, 16), w_packed);
    RF_CHECK_LAUNCH("rf_convv_pack_weight");
    return RF_OK;
}

struct ConvVArgs {
    const float* x;
    const float* wp;
    const float* bias;
    float* out;
    int n, cin, s, cout, cout16, k, stride, so, kpad;
    float slope;
    unsigned gx, gz;     // voxel blocks per window, cout blocks (the grid is 1-D: gx * gz * n workgroups)
};

template <int MB, int NB>
__global__ __launch_bounds__(256) void k_convv_mfma(ConvVArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int so = a.so, s = a.s, cin = a.cin, k = a.k;
    const int ovol = so * so * so;
    const size_t ivol = (size_t)s * s * s;
    // 1-D grid, XCD-aware: workgroups go to the 8 XCDs round-robin by id, each XCD with its own L2.  Every input value of a window is
    // re-read k^3 * (cout blocks) times by different workgroups: remap so that XCD k walks whole windows one after the other
    // (voxel blocks fastest, then cout blocks, then windows) and the window's input stays in that one L2.
    const unsigned total = gridDim.x, per = total >> 3, rem = total & 7u, xk = blockIdx.x & 7u;
    const unsigned lb = xk * per + (xk < rem ? xk : rem) + (blockIdx.x >> 3);
    const unsigned xb = lb % a.gx, zb = (lb / a.gx) % a.gz;
    const int nn = (int)(lb / (a.gx * a.gz));
    const int cob = (int)zb * (NB * 16);
    const int m_wave = ((int)xb * 4 + wave) * (MB * 16);
    if (m_wave >= ovol) return;                                  // whole wave past the end (no barriers in this kernel)

    int base[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        int m = m_wave + mb * 16 + j;
        if (m >= ovol) m = ovol - 1;                             // clamp: computed, never stored
        const int ox = m % so, oy = (m / so) % so, oz = m / (so * so);
        base[mb] = ((oz * a.stride) * s + oy * a.stride) * s + ox * a.stride;
    }
    const float* xin = a.x + (size_t)nn * cin * ivol;
    const float* wl = a.wp + cob + j;

    f32x4 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int K = k * k * k * cin;
    if ((cin & 3) == 0) {
        // cin a multiple of 4: K index = tap*cin + ci walks taps in the outer loop and 4-channel steps inside it -- no integer
        // divisions in the loop (the generic form below spends more issue slots on tap = kidx / cin than on MFMAs)
        const size_t cstep = 4 * ivol;
        const int nc4 = cin >> 2, nsteps = k * k * k * nc4;
        // operands of step i+1 are loaded while step i multiplies (the loads are L1/L2 gathers: latency, not bandwidth)
        int dz = 0, dy = 0, dx = 0, c4 = 0;
        const float* xt = xin + (size_t)kq * ivol;
        const float* wt = wl + (size_t)kq * a.cout16;
        float av[2][MB], bv[2][NB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) av[0][mb] = xt[base[mb]];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bv[0][nb] = (cob + nb * 16 < a.cout16) ? wt[nb * 16] : 0.f;
        for (int i = 0; i < nsteps; i += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (i + h < nsteps) {
                    // advance to step i+h+1: next 4 channels, or the next tap
                    wt += (size_t)4 * a.cout16;
                    if (++c4 == nc4) {
                        c4 = 0;
                        if (++dx == k) { dx = 0; if (++dy == k) { dy = 0; ++dz; } }
                        xt = xin + (size_t)kq * ivol + ((size_t)dz * s + dy) * s + dx;
                    } else {
                        xt += cstep;
                    }
                    if (i + h + 1 < nsteps) {
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) av[h ^ 1][mb] = xt[base[mb]];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) bv[h ^ 1][nb] = (cob + nb * 16 < a.cout16) ? wt[nb * 16] : 0.f;
                    }
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][mb], bv[h][nb], acc[mb][nb], 0, 0, 0);
                }
            }
        }
    } else
    for (int ks = 0; ks < a.kpad; ks += 4) {
        int kidx = ks + kq;
        const bool live = kidx < K;                              // padded k rows carry zero weights; keep the address valid
        if (!live) kidx = 0;
        const int tap = kidx / cin, ci = kidx - tap * cin;
        const int dx = tap % k, dy = (tap / k) % k, dz = tap / (k * k);
        const size_t koff = (size_t)ci * ivol + ((size_t)dz * s + dy) * s + dx;
        float av[MB], bv[NB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) av[mb] = xin[koff + base[mb]];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bv[nb] = (live && cob + nb * 16 < a.cout16) ? wl[(size_t)(ks + kq) * a.cout16 + nb * 16] : 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], bv[nb], acc[mb][nb], 0, 0, 0);
    }

    // epilogue: bias + LeakyReLU; a lane holds 4 consecutive output voxels (linear index) of one cout
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int co = cob + nb * 16 + j;
        if (co >= a.cout) continue;
        const float bz = a.bias ? a.bias[co] : 0.f;
        float* o = a.out + ((size_t)nn * a.cout + co) * ovol;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m_wave + mb * 16 + kq * 4 + r;
                if (m < ovol) {
                    const float v = acc[mb][nb][r] + bz;
                    o[m] = v > 0.f ? v : v * a.slope;
                }
            }
        }
    }
}

extern "C" int rf_conv3d_valid_leaky_mfma(const float* x, int n, int cin, int s, const float* w_packed, const float* bias, int cout, int k,
                                          int stride, float slope, float* out, void* stream) {
    RF_REQUIRE(x && w_packed && out && n > 0 && cin > 0 && cout > 0 && k > 0 && stride > 0 && s >= k, RF_E_INVALID,
               "rf_conv3d_valid_leaky_mfma: bad arguments");
    ConvVArgs a;
    a.x = x; a.wp = w_packed; a.bias = bias; a.out = out;
    a.n = n; a.cin = cin; a.s = s; a.cout = cout; a.cout16 = rf_round_up(cout, 16); a.k = k; a.stride = stride;
    a.so = (s - k) / stride + 1; a.kpad = rf_round_up(k * k * k * cin, 4); a.slope = slope;
    const long long ovol = (long long)a.so * a.so * a.so;
    RF_REQUIRE((long long)cin * s * s * s < (1ll << 31) && ovol < (1ll << 31), RF_E_UNSUPPORTED, "rf_conv3d_valid_leaky_mfma: window too large");
    hipStream_t st = (hipStream_t)stream;
    // 4 waves x MB m-blocks of 16 voxels per workgroup; small outputs take MB = 1 so tiny windows still spread over waves
    if (a.cout16 <= 16) {
        a.gz = 1;
        if (ovol >= 4096) { a.gx = (unsigned)((ovol + 255) / 256); hipLaunchKernelGGL((k_convv_mfma<4, 1>), dim3(a.gx * a.gz * n), dim3(256), 0, st, a); }
        else { a.gx = (unsigned)((ovol + 63) / 64); hipLaunchKernelGGL((k_convv_mfma<1, 1>), dim3(a.gx * a.gz * n), dim3(256), 0, st, a); }
    } else {
        a.gz = (unsigned)((a.cout16 + 31) / 32);
        if (ovol >= 4096) { a.gx = (unsigned)((ovol + 255) / 256); hipLaunchKernelGGL((k_convv_mfma<4, 2>), dim3(a.gx * a.gz * n), dim3(256), 0, st, a); }
        else { a.gx = (unsigned)((ovol + 63) / 64); hipLaunchKernelGGL((k_convv_mfma<1, 2>), dim3(a.gx * a.gz * n), dim3(256), 0, st, a); }
    }
    RF_CHECK_LAUNCH("rf_conv3d_valid_leaky_mfma");
    return RF_OK;
}

// ====================================================================================================================
// LDS-staged form for the LARGE layers of the patch encoders (output edge >= 8: PCPatch48's 1->12 k5 @48^3, 12->24 k3 @44^3,
// 24->48 k3 s2 @42^3, 48->48 k3 s2 @20^3; Patch32's first four layers) -- the gather form above re-reads every input value
// k^3 * cout/16 times through L1/L2 and tops out at 30-50 TFLOP/s on them.
//
//   * workgroup (8 waves x MB 4 m-blocks = 512 output voxels) = tz x ty whole output rows of one window (x runs over the full
//     row, so any edge -- 44, 42, 20, 9 -- tiles without a ragged x border; the host picks (tz, ty) for the least waste) and
//     NB <= 3 cout blocks; M index = linear voxel of the tile, so an m-block may wrap rows: a lane's LDS base is computed once.
//   * K is walked channel by channel in groups of FOUR TAPS (k^3 padded up to a multiple of 4 with zero-weight taps: 27 -> 28,
//     125 -> 128, 8, 64): one MFMA k-step multiplies 4 taps of one channel.  That keeps only CC <= 2 channels of the input tile
//     in LDS at a time (a stride-2 layer's tile is (2 tz + 1)(2 ty + 1) rows: 4 channels would not fit) and makes cin = 1 the same
//     code path.  The 4 lane groups of an A operand read base + tapoff[4 g + kq] (table in LDS).
//   * input rows and the chunk's weight rows [cc][g][kq][cout] are loaded into registers before the MFMA loop of the previous
//     chunk and committed to LDS after it (one LDS buffer, two barriers per chunk); operand reads are double buffered in
//     registers as in conv3d_mfma.hip.
// Weight image: rf_convv_lds_pack_weight -> [cin][G = ceil(k^3/4)][4][cout16].
struct ConvVLArgs {
    const float* x;
    const float* wp;
    const float* bias;
    float* out;
    int n, cin, s, cout, cout16, k, stride, so;
    float slope;
    int tz, ty, ntz, nty, gz;      // output rows per tile, tiles per dim, cout blocks
    int zi, yi, ch;                // staged input rows per channel (z, y extents), LDS floats per channel
    int G;                         // tap groups per channel
    int P, seg;                    // threads per staged input row, floats per thread
};

__global__ void k_convv_lds_pack(const float* __restrict__ w, int cout, int cin, int k3, int G, int cout16, float* __restrict__ wp) {
    const size_t total = (size_t)cin * G * 4 * cout16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % cout16);
        const int tap = (int)((i / cout16) % (4 * G));
        const int ci = (int)(i / ((size_t)cout16 * 4 * G));
        wp[i] = (co < cout && tap < k3) ? w[((size_t)co * cin + ci) * k3 + tap] : 0.f;
    }
}

extern "C" size_t rf_convv_lds_packed_floats(int cout, int cin, int k) {
    return (size_t)cin * ((k * k * k + 3) / 4) * 4 * rf_round_up(cout, 16);
}

extern "C" int rf_convv_lds_pack_weight(const float* w_oidhw, int cout, int cin, int k, float* w_packed, void* stream) {
    RF_REQUIRE(w_oidhw && w_packed && cout > 0 && cin > 0 && k > 0, RF_E_INVALID, "rf_convv_lds_pack_weight: bad arguments");
    const size_t want = (rf_convv_lds_packed_floats(cout, cin, k) + 255) / 256;
    hipLaunchKernelGGL(k_convv_lds_pack, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, (hipStream_t)stream, w_oidhw, cout, cin, k * k * k,
                       (k * k * k + 3) / 4, rf_round_up(cout, 16), w_packed);
    RF_CHECK_LAUNCH("rf_convv_lds_pack_weight");
    return RF_OK;
}

#define RF_VL_SEG 24          // staged floats per thread (one segment of one input row)
template <int NB, int CC>
__global__ __launch_bounds__(512) void k_convv_lds(ConvVLArgs a) {
    constexpr int NT = 512, MB = 4, NCO = NB * 16;
    constexpr int WS = NCO + ((NCO % 32) == 0 ? 16 : 0);           // weight row stride: the 4 k rows of a B read sit on different banks
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kq = lane >> 4;
    const int so = a.so, s = a.s, st = a.stride, G = a.G;
    float* xs = smem;                                               // [CC][ch]
    float* wsl = xs + CC * a.ch;                                    // [CC][G][4][WS]
    int* toff = reinterpret_cast<int*>(wsl + CC * G * 4 * WS);      // [G * 4]

    // XCD-aware 1-D grid as in the gather form: an XCD walks whole windows (tiles fastest, then cout blocks)
    const unsigned total = gridDim.x, per = total >> 3, rem = total & 7u, xk = blockIdx.x & 7u;
    const unsigned lb = xk * per + (xk < rem ? xk : rem) + (blockIdx.x >> 3);
    const unsigned tiles = (unsigned)(a.ntz * a.nty);
    const unsigned tb = lb % tiles, zb = (lb / tiles) % (unsigned)a.gz;
    const int nn = (int)(lb / (tiles * (unsigned)a.gz));
    const int z0 = (int)(tb / (unsigned)a.nty) * a.tz, y0 = (int)(tb % (unsigned)a.nty) * a.ty;
    const int cob = (int)zb * NCO;
    const int V = a.tz * a.ty * so;

    // this lane's output voxels: m-block (wave*MB + mb), voxel j -> LDS base of its (z, y, x) corner
    int base[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        int m = (wave * MB + mb) * 16 + j;
        if (m >= V) m = 0;                                          // computed, never stored
        const int x = m % so, r = m / so, ly = r % a.ty, lz = r / a.ty;
        base[mb] = ((lz * st) * a.yi + ly * st) * s + x * st;
    }
    for (int i = tid; i < G * 4; i += NT) {
        int t = i < a.k * a.k * a.k ? i : 0;                        // zero-weight pad taps read tap 0
        toff[i] = ((t / (a.k * a.k)) * a.yi + (t / a.k) % a.k) * s + t % a.k;
    }

    // ---- staging: one (input row, segment) item per thread
    const size_t ivol = (size_t)s * s * s;
    const float* xin = a.x + (size_t)nn * a.cin * ivol;
    const int rows_c = a.zi * a.yi;                                 // rows per channel
    const int item_row = tid / a.P, item_p = tid % a.P;
    const int it_c = item_row / rows_c, it_r = item_row % rows_c;
    const int it_z = it_r / a.yi, it_y = it_r % a.yi;
    const int iz = z0 * st + it_z, iy = y0 * st + it_y, ix = item_p * a.seg;
    const bool it_inside = iz < s && iy < s;                        // rows past the volume (ragged last tile) are staged as zeros
    int it_cnt = it_c < CC ? s - ix : 0;                            // floats of this thread's segment
    if (it_cnt > a.seg) it_cnt = a.seg;
    if (it_cnt < 0) it_cnt = 0;
    const float* it_src = xin + (size_t)it_c * ivol + ((size_t)iz * s + iy) * s + ix;
    float* it_dst = xs + it_c * a.ch + (it_z * a.yi + it_y) * s + ix;
    float stage[RF_VL_SEG];
    float4 wstage[2];
    const int wrow_f4 = NCO / 4;                                    // float4 per weight row
    const int wtotal = CC * G * 4 * wrow_f4;                        // float4 of a chunk's slab (unpadded rows)

    auto issue = [&](int c0) {
        if (it_inside && c0 + it_c < a.cin) {
            const float* src = it_src + (size_t)c0 * ivol;
#pragma unroll
            for (int i = 0; i < RF_VL_SEG; ++i) if (i < it_cnt) stage[i] = src[i];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int f = tid + h * NT;
            if (f < wtotal) {
                const int row = f / wrow_f4, c4 = f % wrow_f4;      // row = (cc*G + g)*4 + kq
                const int cc = row / (G * 4);
                int co = cob + c4 * 4;
                if (co >= a.cout16) co = 0;                          // block wider than the image: masked at the store
                wstage[h] = (c0 + cc < a.cin) ? *reinterpret_cast<const float4*>(a.wp + ((size_t)c0 * G * 4 + row) * a.cout16 + co)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto commit = [&](int c0) {
        const bool real = it_inside && c0 + it_c < a.cin;
#pragma unroll
        for (int i = 0; i < RF_VL_SEG; ++i) if (i < it_cnt) it_dst[i] = real ? stage[i] : 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int f = tid + h * NT;
            if (f < wtotal) {
                const int row = f / wrow_f4, c4 = f % wrow_f4;
                *reinterpret_cast<float4*>(wsl + row * WS + c4 * 4) = wstage[h];
            }
        }
    };

    f32x4 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue(0);
    commit(0);
    __syncthreads();
    const float* wl = wsl + kq * WS + j;
    for (int c0 = 0; c0 < a.cin; c0 += CC) {
        const bool more = c0 + CC < a.cin;
        if (more) issue(c0 + CC);
        const int ccn = a.cin - c0 < CC ? a.cin - c0 : CC;
        const int nst = ccn * G;                                    // k-steps of this chunk: (cc, g)
        float av[2][MB], bv[2][NB];
        {
            const int to = toff[kq];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) av[0][mb] = xs[base[mb] + to];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bv[0][nb] = wl[nb * 16];
        }
        int g1 = 0, cc1 = 0;                                        // (cc, g) of the step being prefetched
        for (int t = 0; t < nst; t += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (t + h < nst) {
                    asm volatile("" ::: "memory");
                    if (++g1 == G) { g1 = 0; ++cc1; }
                    if (t + h + 1 < nst) {
                        const int to = toff[g1 * 4 + kq];
                        const float* xc = xs + cc1 * a.ch + to;
                        const float* wc = wl + (cc1 * G + g1) * 4 * WS;
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) av[h ^ 1][mb] = xc[base[mb]];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) bv[h ^ 1][nb] = wc[nb * 16];
                    }
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][mb], bv[h][nb], acc[mb][nb], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (more) commit(c0 + CC);
        __syncthreads();
    }

    // ---- epilogue: bias + LeakyReLU, then through LDS so that the stores are long contiguous runs.  The tile is tz planes of
    // ty whole rows: in memory, for one cout and one z, ty * so consecutive floats.  A lane holds 4 consecutive linear voxels of
    // one cout (16-byte LDS write); per cout block the 8 waves then each stream two cout rows out, lane = consecutive voxel.
    // (Direct stores from the accumulator layout are 4-byte pieces scattered over 16 cout planes: the 1->12 k5 and 12->24 k3
    // layers of PCPatch48 write 4.2 / 7.3 GB and were store-bound.)
    constexpr int EV = 512 + 4;                                     // floats per cout row of the epilogue tile
    float* eb = smem;                                               // [16][EV]; the K loop ended on a barrier
    const int ovol = so * so * so;
    const int R = a.ty * so;                                        // floats of one z plane of the tile
    int zlim = so - z0;                                             // valid planes / rows of a ragged last tile
    if (zlim > a.tz) zlim = a.tz;
    int ylim = so - y0;
    if (ylim > a.ty) ylim = a.ty;
    const int mlim = zlim * R, rlim = ylim * so;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        {
            const int co = cob + nb * 16 + j;
            const float bz = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                f32x4 v = acc[mb][nb];
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float t = v[r] + bz; v[r] = t > 0.f ? t : t * a.slope; }
                *reinterpret_cast<f32x4*>(eb + j * EV + (wave * MB + mb) * 16 + kq * 4) = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col = wave * 2 + h, co = cob + nb * 16 + col;
            if (co < a.cout) {                                      // wave-uniform
                float* o = a.out + ((size_t)nn * a.cout + co) * ovol + ((size_t)z0 * so + y0) * so;
                const float* src = eb + col * EV;
                int mr = lane, lz = 0;                              // m = lz * R + mr
                while (mr >= R) { mr -= R; ++lz; }
                for (int m = lane; m < mlim; m += 64) {
                    if (mr < rlim) o[(size_t)lz * so * so + mr] = src[m];
                    mr += 64;
                    while (mr >= R) { mr -= R; ++lz; }
                }
            }
        }
        if (nb + 1 < NB) __syncthreads();
    }
}

// tile choice: (tz, ty) output rows with tz*ty*so <= 512 voxels, <= 512 staged rows per chunk, segments <= RF_VL_SEG floats and
// the chunk within the LDS budget; among those the one that wastes the fewest MFMA slots (ragged last tiles + unfilled m-blocks)
static bool convv_lds_plan(int n, int cin, int s, int cout, int k, int stride, ConvVLArgs& a, int& cc_out, size_t& lds_out) {
    const int so = (s - k) / stride + 1;
    if (so < 8 || k > 5 || s > 64) return false;
    const int cout16 = rf_round_up(cout, 16);
    const int G = (k * k * k + 3) / 4;
    const int nbw = cout16 <= 48 ? cout16 / 16 : (cout16 % 48 == 0 ? 3 : 2);     // cout blocks per workgroup
    const int CC = cin >= 2 ? 2 : 1;
    double best = 0.0;
    for (int tz = 1; tz <= so; ++tz)
        for (int ty = 1; ty <= so; ++ty) {
            const int V = tz * ty * so;
            if (V > 512) continue;
            const int zi = (tz - 1) * stride + k, yi = (ty - 1) * stride + k;
            const int rows = CC * zi * yi;
            if (rows > 512) continue;
            const int P = 512 / rows < 16 ? 512 / rows : 16;
            const int seg = (s + P - 1) / P;
            if (seg > RF_VL_SEG) continue;
            const int ch = zi * yi * s;
            const int ws = nbw * 16 + ((nbw * 16) % 32 == 0 ? 16 : 0);
            size_t lds = ((size_t)CC * ch + (size_t)CC * G * 4 * ws + (size_t)G * 4) * sizeof(float);
            if (lds < (size_t)16 * (512 + 4) * sizeof(float)) lds = (size_t)16 * (512 + 4) * sizeof(float);     // the epilogue tile
            if (lds > 72 * 1024) continue;                          // two workgroups per CU
            if ((size_t)CC * G * 4 * (nbw * 16 / 4) > 1024) continue;   // weight slab: two float4 per thread
            const int ntz = (so + tz - 1) / tz, nty = (so + ty - 1) / ty;
            const double eff = (double)so * so * so / ((double)ntz * nty * 512.0);
            if (eff > best) {
                best = eff;
                a.tz = tz; a.ty = ty; a.ntz = ntz; a.nty = nty; a.zi = zi; a.yi = yi; a.ch = ch; a.P = P; a.seg = seg;
                lds_out = lds;
            }
        }
    if (best < 0.5) return false;
    a.n = n; a.cin = cin; a.s = s; a.cout = cout; a.cout16 = cout16; a.k = k; a.stride = stride; a.so = so; a.G = G;
    a.gz = (cout16 + nbw * 16 - 1) / (nbw * 16);
    cc_out = CC * 10 + nbw;
    return true;
}

extern "C" int rf_conv3d_valid_lds_supported(int n, int cin, int s, int cout, int k, int stride) {
    ConvVLArgs a;
    int v;
    size_t lds;
    return n > 0 && cin > 0 && cout > 0 && k > 0 && stride > 0 && s >= k && convv_lds_plan(n, cin, s, cout, k, stride, a, v, lds) ? 1 : 0;
}

extern "C" int rf_conv3d_valid_leaky_lds(const float* x, int n, int cin, int s, const float* w_packed, const float* bias, int cout, int k,
                                         int stride, float slope, float* out, void* stream) {
    RF_REQUIRE(x && w_packed && out && n > 0 && cin > 0 && cout > 0 && k > 0 && stride > 0 && s >= k, RF_E_INVALID,
               "rf_conv3d_valid_leaky_lds: bad arguments");
    ConvVLArgs a;
    int variant;
    size_t lds;
    RF_REQUIRE(convv_lds_plan(n, cin, s, cout, k, stride, a, variant, lds), RF_E_UNSUPPORTED,
               "rf_conv3d_valid_leaky_lds: shape not taken by the LDS-staged form (ask rf_conv3d_valid_lds_supported; use rf_conv3d_valid_leaky_mfma)");
    a.x = x; a.wp = w_packed; a.bias = bias; a.out = out; a.slope = slope;
    const unsigned grid = (unsigned)a.ntz * a.nty * a.gz * n;
    hipStream_t st = (hipStream_t)stream;
#define RF_VL_LAUNCH(NB_, CC_)                                                                                                   \
    do {                                                                                                                         \
        if (lds > 65536) {                                                                                                       \
            static RfLdsOptIn opt_in;                                                                                            \
            if (int rc = opt_in.ensure(reinterpret_cast<const void*>(k_convv_lds<NB_, CC_>), (int)lds, "rf_conv3d_valid_leaky_lds")) return rc; \
        }                                                                                                                        \
        hipLaunchKernelGGL((k_convv_lds<NB_, CC_>), dim3(grid), dim3(512), lds, st, a);                                          \
    } while (0)
    switch (variant) {
        case 11: RF_VL_LAUNCH(1, 1); break;
        case 12: RF_VL_LAUNCH(2, 1); break;
        case 13: RF_VL_LAUNCH(3, 1); break;
        case 21: RF_VL_LAUNCH(1, 2); break;
        case 22: RF_VL_LAUNCH(2, 2); break;
        default: RF_VL_LAUNCH(3, 2); break;
    }
#undef RF_VL_LAUNCH
    RF_CHECK_LAUNCH("rf_conv3d_valid_leaky_lds");
    return RF_OK;
}

// ====================================================================================================================
// VALU form for the FIRST layers of the patch encoders: stride 1, few channels on both sides (1 -> 8/12 with k = 3/5, 8 -> 16 and
// 12 -> 24 with k = 3).  On the matrix cores these layers pad cout 12 -> 16 / 24 -> 32 (25 % of the MFMA slots) and k^3 to a
// multiple of 4, and stay below 65 TFLOP/s; packed-fp32 VALU has the same 157 TFLOP/s peak and no padding at all.
//   * thread = one (y, x) column of TZ = 4 consecutive output voxels, all COUT accumulators in registers (TZ * COUT <= 96 VGPRs);
//     workgroup = TY whole output rows (TY * so <= 256 columns) x TZ planes of one window;
//   * per input channel and (dy, dx): the column's TZ + K - 1 input values come from the LDS tile (lane = consecutive x); the K
//     weight vectors w[dz][dy][dx][0..COUT) are wave-uniform and come through the SCALAR cache into SGPR pairs (s_load_dwordx4;
//     as LDS broadcasts each of them would cost a full 1-KB data return and the LDS, not the VALU, was the limit), then
//     K * TZ * COUT / 2 v_pk_fma_f32 with an SGPR-pair operand -- every input read feeds up to K * COUT FMAs;
//   * channels are staged in chunks of CC <= 4 (input tile [CC][TZ + K - 1][TY + K - 1][s]); weights: the transposed image
//     [cin][K^3][cout] (a permute of the OIDHW tensor, done by the caller).
struct ConvVVArgs {
    const float* x;
    const float* w;       // [cin][K^3][cout]
    const float* bias;
    float* out;
    int n, cin, cout, s, so, ty, nty, ntz;     // cout: all output channels; a workgroup computes COUT of them (blockIdx.y picks the block)
    float slope;
    int out_pre;          // 1: the output is written in split form for rf_conv3d_valid_leaky_split_ex (conv_valid_split.hip: [4-channel group][h | l][voxel][4 halves])
};

// the column's outputs of one 4-channel group -> bias, LeakyReLU, then the consumer's own scale / clamp / split; lane = consecutive x: 512-byte runs
typedef _Float16 rf_h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void convv_store_split(unsigned char* o, size_t plane_bytes, const float (&v)[4]) {
    rf_h4 hh, ll;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t = __builtin_amdgcn_fmed3f(v[e] * (1.0f / 16), -65504.f, 65504.f);
        const _Float16 h = (_Float16)t;
        hh[e] = h;
        ll[e] = (_Float16)fmaf(-2048.0f, (float)h, t * 2048.0f);
    }
    *reinterpret_cast<rf_h4*>(o) = hh;
    *reinterpret_cast<rf_h4*>(o + plane_bytes) = ll;
}

typedef float rf_v2 __attribute__((ext_vector_type(2)));

// S: the input edge as a compile-time constant for the shipped encoders' layers (0 = run time): the staging index arithmetic divides by
// s / 4, the tile's row count and s -- constants cost a multiply-shift, run-time divisors ~20 VALU instructions each, on the pipe the FMAs need.
template <int COUT, int K, int CC, int S = 0>
__global__ __launch_bounds__(256) void k_convv_valu(ConvVVArgs a) {
    constexpr int TZ = 4, ZI = TZ + K - 1, K3 = K * K * K;
    static_assert(COUT % 4 == 0, "weight vectors are read as float4");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int s = S > 0 ? S : a.s, so = S > 0 ? S - K + 1 : a.so;
    const int ty = S > 0 ? ((256 / (S - K + 1)) < (S - K + 1) ? (256 / (S - K + 1)) : (S - K + 1)) : a.ty;
    const int YI = ty + K - 1;
    float* xs = smem;                                               // [CC][ZI][YI][s]
    const int nty = S > 0 ? (so + ty - 1) / ty : a.nty, ntz = S > 0 ? (so + 3) / 4 : a.ntz;
    const int tiles = ntz * nty;
    const int tb = blockIdx.x % tiles, nn = blockIdx.x / tiles;
    const int cob = blockIdx.y * COUT;
    const int z0 = (tb / nty) * TZ, y0 = (tb % nty) * ty;
    const int ly = tid / so, lx = tid % so;
    const bool col_ok = tid < ty * so && y0 + ly < so;
    const size_t ivol = (size_t)s * s * s;
    const float* xin = a.x + (size_t)nn * a.cin * ivol;
    const int ch = ZI * YI * s;

    rf_v2 acc[TZ][COUT / 2];
#pragma unroll
    for (int z = 0; z < TZ; ++z)
#pragma unroll
        for (int c = 0; c < COUT / 2; ++c) acc[z][c] = (rf_v2){0.f, 0.f};

    for (int c0 = 0; c0 < a.cin; c0 += CC) {
        __syncthreads();                                            // the previous chunk is consumed
        if ((s & 3) == 0) {
            // input rows (full width, 16-byte aligned): float4 pieces, all loads of a thread in flight before the first LDS write
            const int s4 = s >> 2, total4 = CC * ZI * YI * s4;
            constexpr int PASSES = CC == 1 ? 4 : 10;               // >= ceil(total4 / 256) for every shape convv_valu_takes admits (s <= 64)
            float4 st[PASSES];
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int i = tid + p * 256;
                st[p] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < total4) {
                    const int r = i / s4, x4 = i - r * s4;
                    const int yy = r % YI, zz = (r / YI) % ZI, cc = r / (YI * ZI);
                    const int iz = z0 + zz, iy = y0 + yy;
                    if (c0 + cc < a.cin && iz < s && iy < s)
                        st[p] = reinterpret_cast<const float4*>(xin + (size_t)(c0 + cc) * ivol + ((size_t)iz * s + iy) * s)[x4];
                }
            }
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int i = tid + p * 256;
                if (i < total4) reinterpret_cast<float4*>(xs)[i] = st[p];
            }
        } else {
            for (int r = tid; r < CC * ZI * YI; r += 256) {         // odd row lengths: one row per thread and pass
                const int yy = r % YI, zz = (r / YI) % ZI, cc = r / (YI * ZI);
                const int iz = z0 + zz, iy = y0 + yy;
                float* dst = xs + (cc * ZI + zz) * YI * s + yy * s;
                if (c0 + cc < a.cin && iz < s && iy < s) {
                    const float* src = xin + (size_t)(c0 + cc) * ivol + ((size_t)iz * s + iy) * s;
                    for (int i = 0; i < s; ++i) dst[i] = src[i];
                } else {
                    for (int i = 0; i < s; ++i) dst[i] = 0.f;
                }
            }
        }
        __syncthreads();
        if (col_ok) {
            const int ccn = a.cin - c0 < CC ? a.cin - c0 : CC;
            for (int cc = 0; cc < ccn; ++cc) {
#pragma unroll
                for (int dy = 0; dy < K; ++dy)
#pragma unroll
                    for (int dx = 0; dx < K; ++dx) {
                        float col[ZI];
                        const float* xc = xs + cc * ch + (ly + dy) * s + lx + dx;
#pragma unroll
                        for (int i = 0; i < ZI; ++i) col[i] = xc[i * YI * s];
#pragma unroll
                        for (int dz = 0; dz < K; ++dz) {
                            const float* wv = a.w + ((size_t)(c0 + cc) * K3 + (dz * K + dy) * K + dx) * a.cout + cob;    // wave-uniform: scalar loads
#pragma unroll
                            for (int q = 0; q < COUT / 4; ++q) {
                                const rf_v2 wa = {wv[4 * q], wv[4 * q + 1]}, wb = {wv[4 * q + 2], wv[4 * q + 3]};
#pragma unroll
                                for (int z = 0; z < TZ; ++z) {
                                    const rf_v2 xv = {col[z + dz], col[z + dz]};
                                    acc[z][2 * q] = __builtin_elementwise_fma(xv, wa, acc[z][2 * q]);
                                    acc[z][2 * q + 1] = __builtin_elementwise_fma(xv, wb, acc[z][2 * q + 1]);
                                }
                            }
                        }
                    }
            }
        }
    }
    if (!col_ok) return;
    const size_t ovol = (size_t)so * so * so;
    if (a.out_pre) {
#pragma unroll
        for (int g = 0; g < COUT / 4; ++g) {
            float bz[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) bz[e] = a.bias ? a.bias[cob + 4 * g + e] : 0.f;
            unsigned char* o = reinterpret_cast<unsigned char*>(a.out) + (((size_t)nn * (a.cout >> 2) + (cob >> 2) + g) * 2 * ovol + ((size_t)z0 * so + y0 + ly) * so + lx) * 8;
#pragma unroll
            for (int z = 0; z < TZ; ++z)
                if (z0 + z < so) {
                    float v[4] = {acc[z][2 * g][0] + bz[0], acc[z][2 * g][1] + bz[1], acc[z][2 * g + 1][0] + bz[2], acc[z][2 * g + 1][1] + bz[3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.slope;
                    convv_store_split(o + (size_t)z * so * so * 8, ovol * 8, v);
                }
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < COUT / 2; ++c) {
        const float b0 = a.bias ? a.bias[cob + 2 * c] : 0.f, b1 = a.bias ? a.bias[cob + 2 * c + 1] : 0.f;
        float* o0 = a.out + ((size_t)nn * a.cout + cob + 2 * c) * ovol + ((size_t)z0 * so + y0 + ly) * so + lx;
#pragma unroll
        for (int z = 0; z < TZ; ++z) {
            if (z0 + z < so) {
                const float v0 = acc[z][c][0] + b0, v1 = acc[z][c][1] + b1;
                o0[(size_t)z * so * so] = v0 > 0.f ? v0 : v0 * a.slope;
                o0[(size_t)z * so * so + ovol] = v1 > 0.f ? v1 : v1 * a.slope;
            }
        }
    }
}

// The same arithmetic (same FMA order per output: bit-identical results) for single-channel volumes wider than a workgroup's 256 columns --
// the first layer of a patch encoder evaluated ONCE on a whole padded chunk instead of on its 64 overlapping windows (model/retrieval.py
// forward_grid): workgroup = 4 output rows x 64 columns x 4 planes, input tile [4 + K - 1][4 + K - 1][64 + K - 1 rounded up to 4] staged with
// 16-byte loads (s and the tile origin are multiples of 4).
template <int COUT, int K>
__global__ __launch_bounds__(256, 2) void k_convv_valu_xt(ConvVVArgs a) {
    constexpr int TZ = 4, TY = 4, TX = 64, ZI = TZ + K - 1, YI = TY + K - 1, XW = (TX + K - 1 + 3) / 4 * 4, XW4 = XW / 4, K3 = K * K * K;
    static_assert(COUT % 4 == 0, "weight vectors are read as float4");
    __shared__ __attribute__((aligned(16))) float xs[ZI * YI * XW];
    const int tid = threadIdx.x;
    const int s = a.s, so = a.so;
    const int ntx = (so + TX - 1) / TX, nty = (so + TY - 1) / TY, ntz = (so + TZ - 1) / TZ;
    const int tiles = ntz * nty * ntx;
    const int tb = blockIdx.x % tiles, nn = blockIdx.x / tiles;
    const int cob = blockIdx.y * COUT;
    const int x0 = (tb % ntx) * TX, y0 = ((tb / ntx) % nty) * TY, z0 = (tb / (ntx * nty)) * TZ;
    const int ly = tid >> 6, lx = tid & 63;
    const bool col_ok = y0 + ly < so && x0 + lx < so;
    const size_t ivol = (size_t)s * s * s;
    const float* xin = a.x + (size_t)nn * ivol;

    constexpr int TOTAL4 = ZI * YI * XW4, PASSES = (TOTAL4 + 255) / 256;
    {
        float4 st[PASSES];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int i = tid + p * 256;
            st[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < TOTAL4) {
                const int r = i / XW4, x4 = i - r * XW4;
                const int yy = r % YI, zz = r / YI;
                const int iz = z0 + zz, iy = y0 + yy, ix = x0 + x4 * 4;
                if (iz < s && iy < s && ix < s) st[p] = *reinterpret_cast<const float4*>(xin + ((size_t)iz * s + iy) * s + ix);
            }
        }
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int i = tid + p * 256;
            if (i < TOTAL4) reinterpret_cast<float4*>(xs)[i] = st[p];
        }
    }
    __syncthreads();
    if (!col_ok) return;

    rf_v2 acc[TZ][COUT / 2];
#pragma unroll
    for (int z = 0; z < TZ; ++z)
#pragma unroll
        for (int c = 0; c < COUT / 2; ++c) acc[z][c] = (rf_v2){0.f, 0.f};
    // (a loop over the -- one -- input channel, as in k_convv_valu: without it hipcc requests all K^3 weight vectors up front and spills
    // 1,700 SGPRs into lanes of vector registers.  The columns of the next (dy, dx) are read under the FMAs of this one and no further ahead:
    // every LDS address is the thread's base plus a constant, and left alone the scheduler hoists all 25 x 8 reads to the top -- 450 registers)
    for (int cc = 0; cc < a.cin; ++cc) {
        const float* xb = xs + ly * XW + lx;
        float col[2][ZI];
#pragma unroll
        for (int i = 0; i < ZI; ++i) col[0][i] = xb[i * YI * XW];
#pragma unroll
        for (int t = 0; t < K * K; ++t) {
            const int dy = t / K, dx = t % K;
            if (t + 1 < K * K) {
                const float* xc = xb + ((t + 1) / K) * XW + (t + 1) % K;
#pragma unroll
                for (int i = 0; i < ZI; ++i) col[(t + 1) & 1][i] = xc[i * YI * XW];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dz = 0; dz < K; ++dz) {
                const float* wv = a.w + ((size_t)cc * K3 + (dz * K + dy) * K + dx) * a.cout + cob;      // wave-uniform: scalar loads
#pragma unroll
                for (int q = 0; q < COUT / 4; ++q) {
                    const rf_v2 wa = {wv[4 * q], wv[4 * q + 1]}, wb = {wv[4 * q + 2], wv[4 * q + 3]};
#pragma unroll
                    for (int z = 0; z < TZ; ++z) {
                        const rf_v2 xv = {col[t & 1][z + dz], col[t & 1][z + dz]};
                        acc[z][2 * q] = __builtin_elementwise_fma(xv, wa, acc[z][2 * q]);
                        acc[z][2 * q + 1] = __builtin_elementwise_fma(xv, wb, acc[z][2 * q + 1]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const size_t ovol = (size_t)so * so * so;
    if (a.out_pre) {
#pragma unroll
        for (int g = 0; g < COUT / 4; ++g) {
            float bz[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) bz[e] = a.bias ? a.bias[cob + 4 * g + e] : 0.f;
            unsigned char* o = reinterpret_cast<unsigned char*>(a.out) + (((size_t)nn * (a.cout >> 2) + (cob >> 2) + g) * 2 * ovol + ((size_t)z0 * so + y0 + ly) * so + x0 + lx) * 8;
#pragma unroll
            for (int z = 0; z < TZ; ++z)
                if (z0 + z < so) {
                    float v[4] = {acc[z][2 * g][0] + bz[0], acc[z][2 * g][1] + bz[1], acc[z][2 * g + 1][0] + bz[2], acc[z][2 * g + 1][1] + bz[3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.slope;
                    convv_store_split(o + (size_t)z * so * so * 8, ovol * 8, v);
                }
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < COUT / 2; ++c) {
        const float b0 = a.bias ? a.bias[cob + 2 * c] : 0.f, b1 = a.bias ? a.bias[cob + 2 * c + 1] : 0.f;
        float* o0 = a.out + ((size_t)nn * a.cout + cob + 2 * c) * ovol + ((size_t)z0 * so + y0 + ly) * so + x0 + lx;
#pragma unroll
        for (int z = 0; z < TZ; ++z) {
            if (z0 + z < so) {
                const float v0 = acc[z][c][0] + b0, v1 = acc[z][c][1] + b1;
                o0[(size_t)z * so * so] = v0 > 0.f ? v0 : v0 * a.slope;
                o0[(size_t)z * so * so + ovol] = v1 > 0.f ? v1 : v1 * a.slope;
            }
        }
    }
}

// big single-channel volumes (fully-convolutional evaluation of a patch encoder's first layer): the x-tiled kernel
static bool convv_valu_xt_takes(int cin, int s, int cout, int k, int stride) {
    return stride == 1 && cin == 1 && (k == 3 || k == 5) && (cout == 8 || cout == 12 || cout == 16) && s > 64 && s <= 1024 && (s & 3) == 0;
}

static bool convv_valu_takes(int cin, int s, int cout, int k, int stride) {
    if (convv_valu_xt_takes(cin, s, cout, k, stride)) return true;
    if (stride != 1 || (k != 3 && k != 5) || s > 64 || s - k + 1 < 8) return false;
    if (k == 5) return cin == 1 && (cout == 8 || cout == 12);
    return (cin == 1 && (cout == 8 || cout == 12 || cout == 16)) || (cin == 8 && cout == 16) || (cin == 12 && cout == 24);
}

extern "C" int rf_conv3d_valid_valu_supported(int n, int cin, int s, int cout, int k, int stride) {
    return n > 0 && convv_valu_takes(cin, s, cout, k, stride) ? 1 : 0;
}

// x [n][cin][s^3], w_t [cin][k^3][cout] (the OIDHW weight permuted to (1,2,3,4,0)), out [n][cout][so^3]
extern "C" int rf_conv3d_valid_leaky_valu_ex(const float* x, int n, int cin, int s, const float* w_t, const float* bias, int cout, int k,
                                             int stride, float slope, void* out, int out_split, void* stream);
extern "C" int rf_conv3d_valid_leaky_valu(const float* x, int n, int cin, int s, const float* w_t, const float* bias, int cout, int k,
                                          int stride, float slope, float* out, void* stream) {
    return rf_conv3d_valid_leaky_valu_ex(x, n, cin, s, w_t, bias, cout, k, stride, slope, out, 0, stream);
}

// out_split: the output in split form (cout a multiple of 4: every shape this form takes) for rf_conv3d_valid_leaky_split_ex
extern "C" int rf_conv3d_valid_leaky_valu_ex(const float* x, int n, int cin, int s, const float* w_t, const float* bias, int cout, int k,
                                             int stride, float slope, void* out_v, int out_split, void* stream) {
    float* out = reinterpret_cast<float*>(out_v);
    const float* w_oidhw = w_t;
    RF_REQUIRE(x && w_oidhw && out && n > 0, RF_E_INVALID, "rf_conv3d_valid_leaky_valu: bad arguments");
    RF_REQUIRE(convv_valu_takes(cin, s, cout, k, stride), RF_E_UNSUPPORTED,
               "rf_conv3d_valid_leaky_valu: shape not taken by the VALU form (ask rf_conv3d_valid_valu_supported)");
    ConvVVArgs a;
    a.x = x; a.w = w_oidhw; a.bias = bias; a.out = out; a.n = n; a.cin = cin; a.cout = cout; a.s = s; a.so = s - k + 1; a.slope = slope;
    a.out_pre = out_split ? 1 : 0;
    if (convv_valu_xt_takes(cin, s, cout, k, stride)) {
        const size_t tiles = (size_t)((a.so + 3) / 4) * ((a.so + 3) / 4) * ((a.so + 63) / 64) * n;
        RF_REQUIRE(tiles < (1ull << 31), RF_E_INVALID, "rf_conv3d_valid_leaky_valu: too many tiles (%zu)", tiles);
        a.ty = 4; a.nty = (a.so + 3) / 4; a.ntz = (a.so + 3) / 4;
        hipStream_t sx = (hipStream_t)stream;
#define RF_VX(COUT_, K_) hipLaunchKernelGGL((k_convv_valu_xt<COUT_, K_>), dim3((unsigned)tiles), dim3(256), 0, sx, a)
        if (k == 5 && cout == 8) RF_VX(8, 5);
        else if (k == 5 && cout == 12) RF_VX(12, 5);
        else if (k == 5) RF_VX(16, 5);
        else if (cout == 8) RF_VX(8, 3);
        else if (cout == 12) RF_VX(12, 3);
        else RF_VX(16, 3);
#undef RF_VX
        RF_CHECK_LAUNCH("rf_conv3d_valid_leaky_valu");
        return RF_OK;
    }
    a.ty = 256 / a.so;
    if (a.ty > a.so) a.ty = a.so;
    a.nty = (a.so + a.ty - 1) / a.ty;
    a.ntz = (a.so + 3) / 4;
    const int cc = cin >= 4 ? 4 : 1;
    const size_t lds = ((size_t)cc * (4 + k - 1) * (a.ty + k - 1) * s) * sizeof(float);
    RF_REQUIRE(lds <= 64 * 1024, RF_E_UNSUPPORTED, "rf_conv3d_valid_leaky_valu: tile of %zu bytes does not fit LDS", lds);
    const unsigned grid = (unsigned)a.ntz * a.nty * n;
    hipStream_t st = (hipStream_t)stream;
#define RF_VV(COUT_, K_, CC_) hipLaunchKernelGGL((k_convv_valu<COUT_, K_, CC_>), dim3(grid, cout / COUT_), dim3(256), lds, st, a)
#define RF_VVS(COUT_, K_, CC_, S_) hipLaunchKernelGGL((k_convv_valu<COUT_, K_, CC_, S_>), dim3(grid, cout / COUT_), dim3(256), lds, st, a)
    // the shipped encoders' layers with their input edge as a constant (PCPatch48: 48; Patch32: 32 -> 28; model/retrieval.py:4-28,217-243).
    // Not the 12 -> 24 @44 layer: specialised it allocates 234 instead of 196 VGPRs and runs 13.4 -> 22.4 ms.
    if (k == 5 && cout == 12 && s == 48) RF_VVS(12, 5, 1, 48);
    else if (k == 5 && cout == 8 && s == 32) RF_VVS(8, 5, 1, 32);
    else if (k == 3 && cin == 8 && cout == 16 && s == 28) RF_VVS(16, 3, 4, 28);
    else if (k == 5 && cout == 8) RF_VV(8, 5, 1);
    else if (k == 5) RF_VV(12, 5, 1);
    else if (cin == 1 && cout == 8) RF_VV(8, 3, 1);
    else if (cin == 1 && cout == 12) RF_VV(12, 3, 1);
    else if (cin == 1) RF_VV(16, 3, 1);
    else if (cout == 16) RF_VV(16, 3, 4);
    else RF_VV(24, 3, 4);
#undef RF_VV
#undef RF_VVS
    RF_CHECK_LAUNCH("rf_conv3d_valid_leaky_valu");
    return RF_OK;
}
