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
