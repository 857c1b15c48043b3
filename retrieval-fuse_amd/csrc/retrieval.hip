// Online retrieval for gfx950: query windows, exact batched squared-L2 top-k scan over a (sharded) patch database,
// candidate-list merge, same-scene demotion, and the patch gather that composes the retrieved volumes.
//
// Reference code being replaced: util/retrieval.py:79-105 (flann_knn_worker: FLANN nn_index + demotion),
// :145-164 (create_retrieval_from_mapping), dataset/scene.py:61,152-160 and dataset/patched_scene_dataset.py:127-133
// (window cut + normalisation).  FLANN (pyflann, third-party, approximate kd-tree) is NOT restated: the scan below is
// exact.
//
// Scan design (rf_l2_topk)
//   The DB embedding matrix is stored blocked [block][dim][64 rows] so a wave reads one dim of 64 consecutive rows
//   with one coalesced 256-byte load.  A wave keeps its 64 rows x 64 dims in 64 VGPRs (one row per lane) and streams
//   the query tile past them: the query vector comes in through scalar loads (wave-uniform address), each lane
//   accumulates sum (q_d - x_d)^2 in fp32, compares against the list's current k-th best and only on a hit takes the
//   (rare) cooperative insertion path.  Candidates are 64-bit keys (dist_bits << 32 | global_row): for non-negative
//   floats the IEEE bit pattern is monotone, so one unsigned compare orders by (distance, row id) -- ties go to the
//   lower row id, deterministically.  Every (slice, query) gets one sorted list; a bitonic merge kernel reduces
//   them per query.  The same merge serves the RCCL all-gathered per-shard lists.
#include "common.h"

typedef unsigned long long u64;
#define RF_KEY_NONE 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ u64 make_key(float dist, unsigned row) { return ((u64)__float_as_uint(dist) << 32) | row; }

// ---------------------------------------------------------------------------------------------- query windows
// One workgroup per (window, w0) plane, a wave per output ROW (w consecutive elements along the last axis; the element-per-thread form of rounds 1-5
// spent six 64-bit divisions per element -- 0.25 ms for the 191 MB of C5's sixteen 144^3 query grids, 0.19 of the HBM roofline; now the row decode is
// amortised over w elements and a lane costs a compare, a load and the IEEE subtract / divide of the normalisation: same bits)
__global__ __launch_bounds__(256) void k_query_windows(const float* __restrict__ raw, int b, int s, int ps, int ctx, float pad_value,
                                                       float mean, float stddev, float* __restrict__ out) {
    const int np = s / ps, w = ps + 2 * ctx, lane = threadIdx.x & 63;
    // quads of one row never straddle the chunk border when everything is in fours (and raw / out are 16-byte aligned: torch allocations)
    const bool vec4 = ((w | ps | ctx | s) & 3) == 0 && (((size_t)raw | (size_t)out) & 15) == 0;
    // a workgroup = one (window, w0) plane of w rows (decoded once, uniform); its four waves take the rows w1 = wave, wave + 4, ...
    const unsigned plane = blockIdx.x;
    const int w0 = (int)(plane % (unsigned)w);
    const unsigned win = plane / (unsigned)w;
    const int p2 = (int)(win % np), p1 = (int)((win / np) % np), p0 = (int)((win / (np * np)) % np);
    const size_t bb = win / ((unsigned)np * np * np);
    for (int w1 = threadIdx.x >> 6; w1 < w; w1 += 4) {
        const size_t r = (size_t)plane * w + w1;
        const int d0 = p0 * ps + w0 - ctx, d1 = p1 * ps + w1 - ctx, d2b = p2 * ps - ctx;
        const bool row_in = (unsigned)d0 < (unsigned)s && (unsigned)d1 < (unsigned)s;
        const float* src = raw + ((bb * s + (row_in ? d0 : 0)) * s + (row_in ? d1 : 0)) * s;
        float* dst = out + r * w;
        if (vec4) {
            // four consecutive elements per lane (the memory pipes are issue-bound: a quarter of the requests and stores, 16 bytes each); the quad lies in
            // one row, and inside or outside the chunk as a whole
            for (int w2 = 4 * lane; w2 < w; w2 += 256) {
                const int d2 = d2b + w2;
                float4 v = make_float4(pad_value, pad_value, pad_value, pad_value);
                if (row_in && (unsigned)d2 < (unsigned)s) v = *reinterpret_cast<const float4*>(src + d2);
                float4 o;
                o.x = __fdiv_rn(__fsub_rn(v.x, mean), stddev); o.y = __fdiv_rn(__fsub_rn(v.y, mean), stddev);
                o.z = __fdiv_rn(__fsub_rn(v.z, mean), stddev); o.w = __fdiv_rn(__fsub_rn(v.w, mean), stddev);
                *reinterpret_cast<float4*>(dst + w2) = o;
            }
        } else {
            for (int w2 = lane; w2 < w; w2 += 64) {
                const int d2 = d2b + w2;
                float v = pad_value;
                if (row_in && (unsigned)d2 < (unsigned)s) v = src[d2];
                dst[w2] = __fdiv_rn(__fsub_rn(v, mean), stddev);
            }
        }
    }
}

extern "C" int rf_query_windows(const float* raw, int b, int s, int ps, int ctx, float pad_value, float mean, float stddev,
                                float* out, void* stream) {
    RF_REQUIRE(raw && out && b > 0 && s > 0 && ps > 0 && ctx >= 0 && s % ps == 0, RF_E_INVALID, "rf_query_windows: bad arguments");
    const int np = s / ps, w = ps + 2 * ctx;
    const size_t planes = (size_t)b * np * np * np * w;
    RF_REQUIRE(planes < (1ull << 31), RF_E_INVALID, "rf_query_windows: too many window planes (%zu)", planes);
    hipLaunchKernelGGL(k_query_windows, dim3((unsigned)planes), dim3(256), 0, (hipStream_t)stream, raw, b, s, ps, ctx, pad_value, mean, stddev, out);
    RF_CHECK_LAUNCH("rf_query_windows");
    return RF_OK;
}

// ---------------------------------------------------------------------------------------- windows of a feature grid
// The patch encoders evaluated fully convolutionally (model/retrieval.py forward_grid): the first layers run once on the padded chunk, then
// the (np)^3 windows of edge w at stride `step` are cut out of the feature grid [n][c][g^3] -> [(n np^3)][c][w^3] for the remaining layers.
// One workgroup per (window, channel), a wave per w0 plane of w x w elements (a lane's (w1, w2) by one float reciprocal): the element-per-thread form of rounds 3-5
// decoded every element with 64-bit divisions.
template <typename T>
__global__ __launch_bounds__(256) void k_gather_windows(const T* __restrict__ grid, int n, int c, int g, int w, int step, int np,
                                                        T* __restrict__ out) {
    const int lane = threadIdx.x & 63, w2n = w * w;
    const float inv_w = 1.0f / (float)w;
    const size_t g3 = (size_t)g * g * g;
    // a workgroup = one (window, channel) (decoded once, uniform); its four waves take the planes w0 = wave, wave + 4, ...
    const unsigned cw = blockIdx.x;
    const int ch = (int)(cw % (unsigned)c);
    const unsigned win = cw / (unsigned)c;
    const int p2 = (int)(win % np), p1 = (int)((win / np) % np), p0 = (int)((win / (np * np)) % np);
    const size_t bb = win / ((unsigned)np * np * np);
    const T* src0 = grid + (bb * c + ch) * g3 + ((size_t)(p0 * step) * g + p1 * step) * g + p2 * step;
    T* dst0 = out + (size_t)cw * w * w2n;
    for (int w0 = threadIdx.x >> 6; w0 < w; w0 += 4) {
        const T* src = src0 + (size_t)w0 * g * g;
        T* dst = dst0 + (size_t)w0 * w2n;
        for (int e = lane; e < w2n; e += 64) {
            const int w1 = (int)(((float)e + 0.5f) * inv_w), w2 = e - w1 * w;      // e < 2^16: (e + 0.5) / w truncates to e / w
            dst[e] = src[(size_t)w1 * g + w2];
        }
    }
}

extern "C" int rf_gather_windows(const float* grid, int n, int c, int g, int w, int step, int np, float* out, void* stream) {
    RF_REQUIRE(grid && out && n > 0 && c > 0 && g > 0 && w > 0 && step > 0 && np > 0 && (np - 1) * step + w <= g, RF_E_INVALID,
               "rf_gather_windows: bad arguments (the last window must end inside the grid)");
    RF_REQUIRE(w <= 255, RF_E_INVALID, "rf_gather_windows: window edge %d", w);
    const size_t blocks = (size_t)n * np * np * np * c;
    RF_REQUIRE(blocks < (1ull << 31), RF_E_INVALID, "rf_gather_windows: too many (window, channel) pairs (%zu)", blocks);
    hipLaunchKernelGGL(k_gather_windows<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grid, n, c, g, w, step, np, out);
    RF_CHECK_LAUNCH("rf_gather_windows");
    return RF_OK;
}

// the same cut on a tensor in the valid convs' split form ([n][c/4][h | l][g^3] 8-byte slots): 2 c/4 planes of 8-byte voxels per sample
extern "C" int rf_gather_windows_split(const void* grid, int n, int c, int g, int w, int step, int np, void* out, void* stream) {
    RF_REQUIRE(grid && out && n > 0 && c > 0 && (c & 3) == 0 && g > 0 && w > 0 && step > 0 && np > 0 && (np - 1) * step + w <= g, RF_E_INVALID,
               "rf_gather_windows_split: bad arguments (channels in fours; the last window must end inside the grid)");
    RF_REQUIRE(w <= 255, RF_E_INVALID, "rf_gather_windows_split: window edge %d", w);
    const size_t blocks = (size_t)n * np * np * np * (c / 2);
    RF_REQUIRE(blocks < (1ull << 31), RF_E_INVALID, "rf_gather_windows_split: too many (window, channel pair) units (%zu)", blocks);
    hipLaunchKernelGGL(k_gather_windows<double>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const double*>(grid), n, c / 2, g, w, step, np, reinterpret_cast<double*>(out));
    RF_CHECK_LAUNCH("rf_gather_windows_split");
    return RF_OK;
}

// ------------------------------------------------------------------------------------------------- DB packing
// The packed image holds three views of the shard's embedding matrix (dim = 64):
//   blocked  [ceil(n/64)][64 dims][64 rows]      the VALU scan: one coalesced 256-byte load per dim per wave
//   rows     [n32][64], n32 = n rounded up to 32   the MFMA scan's A operand: position 16*g + m of a row holds dim 4*m + g, so
//                                                the 64 contiguous bytes lane (row, g) loads are the dims {4m + g} = the g-th
//                                                summation chain of the exact distance (see rf_exact_dist below)
//   hd       [n32]                               (1 - 2^-15) * |row|^2 / 2, +inf for the padding rows (they never pass the filter)
//   rows16   [n32][64] f16                       the rows rounded to f16, natural dim order: A operands of the f16-filtered scan (the h pieces)
//   hd16     [n32]                               (1 - 2^-15) * |row|^2 / 2; -inf for a row with a component beyond the f16 range (always re-checked)
//   rows16l  [n32][64] f16                       the l pieces: (x - h) * 2^11 rounded to f16 -- x = h + l / 2^11 up to 2^-22 |x| (round 6: the filter multiplies
//                                                split operands, so that it is as tight as the fp32-MFMA filter at a fifth of its matrix-pipe cycles)
#define RF_DIM 64
#define RF_EPS_FILTER 3.0517578125e-05f            // 2^-15, see the error bound at k_l2_topk_mfma

static inline size_t rf_blocked_floats(int64_t n) { return (size_t)((n + 63) / 64) * RF_DIM * 64; }
static inline int64_t rf_rows32(int64_t n) { return (n + 31) / 32 * 32; }

__global__ __launch_bounds__(256) void k_db_pack(const float* __restrict__ emb, long long n, float* __restrict__ blocked, float* __restrict__ rows,
                                                 float* __restrict__ hd, _Float16* __restrict__ rows16, float* __restrict__ hd16, _Float16* __restrict__ rows16l) {
    const long long nblk = (n + 63) / 64, n32 = (n + 31) / 32 * 32;
    const size_t total = (size_t)nblk * RF_DIM * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i % 64), d = (int)((i / 64) % RF_DIM);
        const long long row = (long long)(i / ((size_t)64 * RF_DIM)) * 64 + r;
        blocked[i] = row < n ? emb[(size_t)row * RF_DIM + d] : 0.f;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n32 * RF_DIM; i += (size_t)gridDim.x * blockDim.x) {
        const long long row = (long long)(i / RF_DIM);
        const int p = (int)(i % RF_DIM), g = p >> 4, m = p & 15;
        rows[i] = row < n ? emb[(size_t)row * RF_DIM + 4 * m + g] : 0.f;
        {   // a row with a component beyond the f16 range is all zeros here and has hd16 = -inf: it passes the filter and is re-checked exactly
            bool fits = row < n;
            if (fits)
                for (int d = 0; d < RF_DIM; ++d) fits = fits && fabsf(emb[(size_t)row * RF_DIM + d]) < 6.0e4f;
            const float xv = fits ? emb[(size_t)row * RF_DIM + p] : 0.f;
            const _Float16 xh = (_Float16)xv;
            rows16[i] = xh;
            rows16l[i] = (_Float16)((xv - (float)xh) * 2048.f);      // (x - h is exact in fp32)
        }
    }
    for (long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x; row < n32; row += (long long)gridDim.x * blockDim.x) {
        float h = INFINITY, h16 = INFINITY;
        if (row < n) {
            double nd = 0.0, big = 0.0;
            for (int d = 0; d < RF_DIM; ++d) { const double v = emb[(size_t)row * RF_DIM + d]; nd += v * v; big = fmax(big, fabs(v)); }
            h = (float)((1.0 - (double)RF_EPS_FILTER) * 0.5 * nd);
            h16 = big < 6.0e4 ? (float)((1.0 - (double)RF_EPS_FILTER) * 0.5 * nd) : -INFINITY;
        }
        hd[row] = h;
        hd16[row] = h16;
    }
}

extern "C" size_t rf_db_packed_floats(int64_t n, int dim) {
    (void)dim;
    return rf_blocked_floats(n) + (size_t)rf_rows32(n) * RF_DIM + (size_t)rf_rows32(n) + (size_t)rf_rows32(n) * (RF_DIM / 2) + (size_t)rf_rows32(n) + (size_t)rf_rows32(n) * (RF_DIM / 2);
}

extern "C" int rf_db_pack_embeddings(const float* emb, int64_t n, int dim, float* packed, void* stream) {
    RF_REQUIRE(emb && packed && n > 0, RF_E_INVALID, "rf_db_pack_embeddings: bad arguments");
    RF_REQUIRE(dim == RF_DIM, RF_E_UNSUPPORTED, "rf_db_pack_embeddings: embedding dim %d (only 64, the latent_dim of every shipped config)", dim);
    float* rows = packed + rf_blocked_floats(n);
    float* hd = rows + (size_t)rf_rows32(n) * RF_DIM;
    _Float16* rows16 = reinterpret_cast<_Float16*>(hd + rf_rows32(n));
    float* hd16 = reinterpret_cast<float*>(rows16 + (size_t)rf_rows32(n) * RF_DIM);
    _Float16* rows16l = reinterpret_cast<_Float16*>(hd16 + rf_rows32(n));
    const size_t want = (rf_blocked_floats(n) + 255) / 256;
    hipLaunchKernelGGL(k_db_pack, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, (hipStream_t)stream, emb, (long long)n, packed, rows, hd, rows16, hd16, rows16l);
    RF_CHECK_LAUNCH("rf_db_pack_embeddings");
    return RF_OK;
}

// --------------------------------------------------------------------------------------------------- the scans
// THE exact distance of the path (both scans, every code path): four fp32 FMA chains, chain c over the dims d = 4m + c in
// the order m = 0..15 with t = q_d - x_d (rounded), acc = fma(t, t, acc) from 0; dist = (c0 + c2) + (c1 + c3).
// The VALU scan runs the chains as two packed-fp32 accumulators {c0,c1}, {c2,c3} over dim pairs (4m, 4m+1), (4m+2, 4m+3);
// the MFMA scan's re-check runs chain g on the lane that holds row chunk g.  Same bits everywhere, so a (query, row) pair has
// ONE distance no matter which scan, slice or shard evaluates it.
#define RF_TQ 64          // queries per workgroup tile of the VALU scan

// cooperative sorted insert of one wave-uniform candidate into the list held by lanes 0..K2-1 (ascending keys)
template <int K2>
__device__ __forceinline__ void list_insert(u64& e, int lane, u64 cand) {
    const unsigned long long less = __ballot(e < cand) & ((1ull << K2) - 1ull);
    const int pos = __popcll(less);                              // entries smaller than cand form a prefix
    const u64 up = __shfl_up(e, 1, 64);
    if (lane < K2) e = lane < pos ? e : (lane == pos ? cand : up);
}

// ascending bitonic sort of one key per lane across the wave (64 keys, 21 compare-exchange steps on the crossbar)
__device__ __forceinline__ u64 wave_sort64(u64 key, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const u64 other = __shfl_xor(key, j, 64);
            const bool asc = (lane & k) == 0;
            const bool lower = (lane & j) == 0;
            const u64 mn = key < other ? key : other, mx = key < other ? other : key;
            key = (lower == asc) ? mn : mx;
        }
    }
    return key;
}

// VALU scan.  One workgroup = (DB slice, tile of RF_TQ queries); its 4 waves SPLIT THE QUERIES (RF_QW each) and every wave walks
// all 64-row blocks of the slice, so there is exactly one candidate list per (slice, query).  A wave's lists live in
// registers (list j: entry i in lane i of e[j]).  The first block initialises a list with a wave-wide sort; afterwards a
// row enters only if it beats the list's current worst (ballot), which becomes rare quickly (~K2/b hits for block b).
// Every pair costs 64 packed VALU instructions: the right tool for small shards (<= ~100 k rows per GPU).
#define RF_QW (RF_TQ / 4)
typedef float rf_f32x2 __attribute__((ext_vector_type(2)));
template <int K2>
__global__ __launch_bounds__(256) void k_l2_topk(const float* __restrict__ q, int nq, const float* __restrict__ db, long long n,
                                                 unsigned row_base, int blocks_per_slice, u64* __restrict__ parts) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slice = blockIdx.x, q0 = blockIdx.y * RF_TQ + wave * RF_QW;
    const long long nblk = (n + 63) / 64;
    const long long blk_lo = (long long)slice * blocks_per_slice;
    long long blk_hi = blk_lo + blocks_per_slice;
    if (blk_hi > nblk) blk_hi = nblk;

    u64 e[RF_QW];
#pragma unroll
    for (int j = 0; j < RF_QW; ++j) e[j] = RF_KEY_NONE;

    for (long long blk = blk_lo; blk < blk_hi; ++blk) {
        // this lane's DB row: 64 coalesced loads, one per dim (the 4 waves read the same block: L1/L2 hits), kept as register
        // PAIRS of consecutive dims: the distance loop runs on the packed-fp32 VALU (v_pk_add_f32 / v_pk_fma_f32), queries as
        // scalar-register pairs
        rf_f32x2 x[RF_DIM / 2];
        const float* bp = db + (size_t)blk * RF_DIM * 64 + lane;
#pragma unroll
        for (int d = 0; d < RF_DIM / 2; ++d) x[d] = (rf_f32x2){bp[(2 * d) * 64], bp[(2 * d + 1) * 64]};
        const long long row = blk * 64 + lane;
        const bool valid = row < n;
        const unsigned grow = row_base + (unsigned)row;
        const bool first = blk == blk_lo;

#pragma unroll
        for (int j = 0; j < RF_QW; ++j) {
            const int qi = q0 + j;
            if (qi < nq) {                                           // wave-uniform
                const float* qp = q + (size_t)qi * RF_DIM;           // wave-uniform address -> scalar loads
                rf_f32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};     // chains {c0, c1} and {c2, c3} (see "THE exact distance")
#pragma unroll
                for (int m = 0; m < RF_DIM / 4; ++m) {
                    const rf_f32x2 qa = {qp[4 * m], qp[4 * m + 1]}, qb = {qp[4 * m + 2], qp[4 * m + 3]};
                    const rf_f32x2 ta = qa - x[2 * m], tb = qb - x[2 * m + 1];
                    acc01 = __builtin_elementwise_fma(ta, ta, acc01);
                    acc23 = __builtin_elementwise_fma(tb, tb, acc23);
                }
                // (c0 + c2) + (c1 + c3), the last add as an explicit v_add_f32: written as `sum[0] + sum[1]` hipcc forms it as
                // v_pk_add_f32 sum, sum op_sel:[0,1] op_sel_hi:[1,0] -- a packed form that reads zero in lanes 48..63 while another wave's
                // F16 MFMA runs on the SIMD (build.py:unsafe_packed_fp32)
                const rf_f32x2 sum = acc01 + acc23;                  // {c0 + c2, c1 + c3}
                float dist;
                asm("v_add_f32 %0, %1, %2" : "=v"(dist) : "v"(sum[0]), "v"(sum[1]));
                const u64 key = valid ? make_key(dist, grow) : RF_KEY_NONE;
                if (first) {
                    const u64 sorted = wave_sort64(key, lane);
                    e[j] = lane < K2 ? sorted : RF_KEY_NONE;
                } else {
                    const unsigned wlo = __builtin_amdgcn_readlane((unsigned)(e[j] & 0xffffffffu), K2 - 1);
                    const unsigned whi = __builtin_amdgcn_readlane((unsigned)(e[j] >> 32), K2 - 1);
                    const u64 worst = ((u64)whi << 32) | wlo;
                    unsigned long long hits = __ballot(key < worst);
                    while (hits) {
                        const int src = __ffsll((long long)hits) - 1;
                        hits &= hits - 1;
                        const unsigned lo = __builtin_amdgcn_readlane((unsigned)(key & 0xffffffffu), src);
                        const unsigned hi = __builtin_amdgcn_readlane((unsigned)(key >> 32), src);
                        list_insert<K2>(e[j], lane, ((u64)hi << 32) | lo);
                    }
                }
            }
        }
    }
    // publish: parts[slice][q][K2]
#pragma unroll
    for (int j = 0; j < RF_QW; ++j) {
        const int qi = q0 + j;
        if (qi < nq && lane < K2) parts[((size_t)slice * nq + qi) * K2 + lane] = e[j];
    }
}

// MFMA-filtered scan for large shards.  q.x for a 32-row x 64-query tile costs 128 v_mfma_f32_16x16x4_f32 (the VALU scan spends
// 2 x 64 packed instructions PER PAIR); the dot product only FILTERS -- whatever survives is re-evaluated with THE exact
// distance and inserted by key, so the lists are bit-identical to the VALU scan's:
//   * wave = 64 queries (4 n-blocks), stationary in 64 VGPRs as B operands (lane (n, g) holds dims {4j + g} of query n);
//     DB rows stream through as A operands, 2 m-blocks of 16 rows per step, lane (r, g) loading the 64 contiguous bytes of
//     row r's chunk g from the `rows` image (double buffered in registers); 4 waves of a workgroup = 256 queries on the same
//     slice (L1/L2 hits), no LDS staging, no barrier anywhere.
//   * the accumulators start at -hd[row], so after the 16 k-steps  s' = q.x - (1-eps)|x|^2/2.  With T the current k2-th best
//     EXACT distance of the query's list:  |q-x|^2 < T  =>  s' >= a_q := (1-eps)|q|^2/2 - T/2   (eps = 2^-15; proof below).
//     One v_cmp per accumulator register; a step without a passing pair (almost all of them) costs nothing else.
//   * a passing pair (~k2 * ln(rows per slice / k2) per query and slice) is re-evaluated from the registers that already hold
//     it: the query chunk comes over by ds_bpermute, chain g runs on lane (r, g), four readlanes combine.  Then the usual
//     cooperative sorted insert into the wave-private list in LDS and a_q is refreshed.
//   * the lists start EMPTY with a threshold from a sample pass: rf_l2_topk first runs the VALU scan over the shard's first
//     rows (n/64, 1 k..16 k of them); T0[q] = the sample's k2-th best exact distance bounds the final k2-th distance from above,
//     so only pairs with distance <= T0 can matter (the scan covers the sample rows again).  Without it every one of the up to 64
//     slices would re-discover the threshold: ~k2 ln(rows per slice / k2) re-checks per query and slice instead of
//     ~k2 * n / sample per query over the whole shard (1 M rows, 2048 queries: 3840 -> 490 re-checks per query).
// Error bound (why nothing is missed): for fp32 chains of length <= 66 the computed s' differs from the real q.x - hd by at
// most 66*2^-24 * (sum|q_d x_d| + hd) <= 3.0e-6 (|q|^2+|x|^2); THE exact distance differs from the real |q-x|^2 by at most
// 66*2^-24 |q-x|^2 <= 7.9e-6 (|q|^2+|x|^2); hd and a_q add rounding of 2^-24.  Real |q-x|^2 = |q|^2 + |x|^2 - 2 q.x, so a pair
// whose exact distance is below T has  s' >= (|q|^2+|x|^2)(1/2 - 7.0e-6) - T/2 - (1-eps)|x|^2/2 >= a_q  as soon as
// eps/2 >= 7.0e-6 + 3.0e-6, i.e. eps >= 2.0e-5; eps = 2^-15 = 3.05e-5.  (Rows are visited in ascending id order, so a pair
// that TIES the list's worst distance can never enter: "<" is enough.)
template <int K2>
__global__ __launch_bounds__(256, 2) void k_l2_topk_mfma(const float* __restrict__ q, int nq, const float* __restrict__ rows_img,
                                                         const float* __restrict__ hd, long long n, unsigned row_base, int rows_per_slice,
                                                         const float* __restrict__ t0, int t0_stride, u64* __restrict__ parts) {
    __shared__ u64 s_lists[4][64 * K2];
    __shared__ float s_aq[4][64], s_hq[4][64], s_t0[4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slice = blockIdx.x;
    const int q0 = (blockIdx.y * 4 + wave) * 64;                     // this wave's 64 queries
    if (q0 >= nq) return;                                            // no barrier in this kernel: a wave may leave
    const long long n32 = (n + 31) / 32 * 32;
    const long long r_lo = (long long)slice * rows_per_slice;        // multiple of 64
    long long r_hi = r_lo + rows_per_slice;
    if (r_hi > n32) r_hi = n32;
    u64* lists = s_lists[wave];
    float* aqs = s_aq[wave];
    float* hqs = s_hq[wave];
    float* t0s = s_t0[wave];
    const int li = lane & 15, lg = lane >> 4;

    // ---------------------------------------------------------------- lists start empty, thresholds from the sample pass
    // t0[q] = k2-th best exact distance among the shard's first rows (rf_l2_topk's sample pass, VALU scan): an upper bound of
    // the final k2-th distance (+inf when the sample holds fewer than k2 rows).
    for (int ql = lane; ql < 64; ql += 64) {
        const int qi = q0 + ql;
        float hq = 0.f, a = INFINITY;                                // a query that does not exist: nothing passes
        if (qi < nq) {
            const float* qp = q + (size_t)qi * RF_DIM;
            float nqn = 0.f;
#pragma unroll
            for (int d = 0; d < RF_DIM; ++d) nqn = fmaf(qp[d], qp[d], nqn);
            hq = (1.f - RF_EPS_FILTER) * 0.5f * nqn;
            const float t = t0[(size_t)qi * t0_stride];
            a = hq - 0.5f * t;                                       // t = +inf (sample smaller than k2): -inf, everything passes
        }
        hqs[ql] = hq;
        aqs[ql] = a;
        t0s[ql] = qi < nq ? t0[(size_t)qi * t0_stride] : 0.f;
    }
    for (int i = lane; i < 64 * K2; i += 64) lists[i] = RF_KEY_NONE;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---------------------------------------------------------------- the filtered scan
    float b[4][16];                                                  // B operands: b[nb][j] = dim 4j + g of query q0 + nb*16 + n
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int qi = q0 + nb * 16 + li;
#pragma unroll
        for (int j = 0; j < 16; ++j) b[nb][j] = qi < nq ? q[(size_t)qi * RF_DIM + 4 * j + lg] : 0.f;
    }
    float aq[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) aq[nb] = aqs[nb * 16 + li];

    auto load_tile = [&](long long row0, float (&a)[2][16], f32x4 (&h)[2]) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const float4* rp = reinterpret_cast<const float4*>(rows_img + (size_t)(row0 + mb * 16 + li) * RF_DIM + 16 * lg);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float4 t = rp[k]; a[mb][4 * k] = t.x; a[mb][4 * k + 1] = t.y; a[mb][4 * k + 2] = t.z; a[mb][4 * k + 3] = t.w; }
            const float4 t = *reinterpret_cast<const float4*>(hd + row0 + mb * 16 + 4 * lg);
            h[mb] = (f32x4){-t.x, -t.y, -t.z, -t.w};
        }
    };

    auto scan_tile = [&](long long row0, const float (&a)[2][16], const f32x4 (&h)[2]) {
        f32x4 acc[2][4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = h[mb];
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb][j], b[nb][j], acc[mb][nb], 0, 0, 0);
        // does ANY pair of the tile pass?  per accumulator register one v_cmp whose lane mask is OR-ed on the scalar unit
        unsigned long long anyhit = 0ull;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) anyhit |= __ballot(acc[mb][nb][i] >= aq[nb]);
        if (anyhit == 0ull) return;
        unsigned m = 0u;                                              // bit (mb*4 + nb)*4 + i: D row 4*lg + i of m-block mb, query column li of n-block nb
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) m |= (acc[mb][nb][i] >= aq[nb] ? 1u : 0u) << ((mb * 4 + nb) * 4 + i);
        // ---- some pair passed the filter: exact re-check from the registers that hold it
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const unsigned m4 = (m >> ((mb * 4 + nb) * 4)) & 15u;
                unsigned long long bal = __ballot(m4 != 0u);
                while (bal) {
                    const int src = __ffsll((long long)bal) - 1;     // lane (n_, g_) of the D tile
                    bal &= bal - 1;
                    unsigned mi = __builtin_amdgcn_readlane(m4, src);
                    const int n_ = src & 15, g_ = src >> 4;
                    const int ql = nb * 16 + n_;
                    while (mi) {
                        const int i = __ffs((int)mi) - 1;
                        mi &= mi - 1;
                        const int r = 4 * g_ + i;                     // row of the m-block
                        const long long row = row0 + mb * 16 + r;
                        if (row >= n) continue;
                        // chain g of THE exact distance on lane (r, g): the query chunk comes from lane (n_, g)
                        float P = 0.f;
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const float qv = __shfl(b[nb][j], n_ + (lane & 48), 64);
                            const float t = qv - a[mb][j];
                            P = fmaf(t, t, P);
                        }
                        const unsigned Pb = __float_as_uint(P);        // readlane moves 32-bit patterns
                        const float c0 = __uint_as_float(__builtin_amdgcn_readlane(Pb, r)), c1 = __uint_as_float(__builtin_amdgcn_readlane(Pb, r + 16));
                        const float c2 = __uint_as_float(__builtin_amdgcn_readlane(Pb, r + 32)), c3 = __uint_as_float(__builtin_amdgcn_readlane(Pb, r + 48));
                        const float dist = (c0 + c2) + (c1 + c3);
                        const u64 key = make_key(dist, row_base + (unsigned)row);
                        u64 e = lane < K2 ? lists[ql * K2 + lane] : RF_KEY_NONE;
                        const unsigned wlo = __builtin_amdgcn_readlane((unsigned)(e & 0xffffffffu), K2 - 1);
                        const unsigned whi = __builtin_amdgcn_readlane((unsigned)(e >> 32), K2 - 1);
                        if (key < (((u64)whi << 32) | wlo) && dist <= t0s[ql]) {          // (<=: the sample's own k2-th row must get in)
                            list_insert<K2>(e, lane, key);
                            if (lane < K2) lists[ql * K2 + lane] = e;
                            if (lane == K2 - 1) {                    // T = min(sample bound, the list's k2-th distance once it is full)
                                const float tl = e == RF_KEY_NONE ? INFINITY : __uint_as_float((unsigned)(e >> 32));
                                aqs[ql] = hqs[ql] - 0.5f * fminf(tl, t0s[ql]);
                            }
                        }
                    }
                }
            }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) aq[nb] = aqs[nb * 16 + li];
    };

    float a0[2][16], a1[2][16];
    f32x4 h0[2], h1[2];
    long long row0 = r_lo;
    if (row0 < r_hi) load_tile(row0, a0, h0);
    while (row0 < r_hi) {
        if (row0 + 32 < r_hi) load_tile(row0 + 32, a1, h1);
        scan_tile(row0, a0, h0);
        row0 += 32;
        if (row0 >= r_hi) break;
        if (row0 + 32 < r_hi) load_tile(row0 + 32, a0, h0);
        scan_tile(row0, a1, h1);
        row0 += 32;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // publish: parts[slice][q][K2]
    for (int ql = 0; ql < 64; ++ql) {
        const int qi = q0 + ql;
        if (qi < nq && lane < K2) parts[((size_t)slice * nq + qi) * K2 + lane] = lists[ql * K2 + lane];
    }
}

// The same scan with the FILTER on the F16 matrix cores, on SPLIT operands (round 6): x = h + l / 2^11 with h = f16(x), l = f16((x - h) 2^11), rows and queries
// alike, and  q.x ~ qh.xh + (qh.xl + ql.xh) / 2^11  -- 6 x v_mfma_f32_16x16x32_f16 per 16 x 16 tile instead of 16 x v_mfma_f32_16x16x4_f32 (a fifth of the
// matrix-pipe cycles), exact products, separate fp32 accumulators for the two orders of magnitude.  What is dropped: ql.xl / 2^22 and the f16 rounding of the
// l pieces, together <= 3 * 2^-22 sum|q_d x_d| <= 3.6e-7 (|q|^2 + |x|^2) -- an eighth of the fp32 chains' own accumulation error -- so the bound of
// k_l2_topk_mfma carries over with eps/2 >= 7.0e-6 + 3.4e-6: eps = 2^-15 as there.
// Rounds 3-5 multiplied the h pieces alone (2 MFMAs per tile, eps = 2^-9): a slack of 4e-3 in the squared distance.  On isotropic embeddings (squared distances 2
// +- 0.25) that is nothing; on CLUSTERED ones -- a database that lies where the queries lie, squared distances to the nearest rows 5e-3..2e-2, which is what
// a trained encoder pair produces and what bench.py builds since round 6 -- every row of the query's neighbourhood passed such a filter and was re-checked one by
// one: 2048 queries against 1 M rows 1.1 ms -> 15.4 ms.  Rows or queries with a component beyond the f16 range get hd = -inf / a_q = -inf (always re-checked).
// Pairs that pass are re-evaluated with THE exact distance from the fp32 rows (fetched for that tile only), so the lists are bit-identical to the other scans'.
#define RF_EPS_FILTER16 RF_EPS_FILTER              // 2^-15
typedef _Float16 rf_h8 __attribute__((ext_vector_type(8)));
// [query][chain g][16] = q[query][4 j + g]: the four summation chains of THE exact distance, contiguous per chain (k_l2_topk_mfma16's re-check reads them)
__global__ __launch_bounds__(256) void k_query_chains(const float* __restrict__ q, int nq, float* __restrict__ qc) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)nq * RF_DIM) return;
    const int p = (int)(i % RF_DIM), g = p >> 4, j = p & 15;
    qc[i] = q[(i / RF_DIM) * RF_DIM + 4 * j + g];
}

template <int K2>
__global__ __launch_bounds__(256, 2) void k_l2_topk_mfma16(const float* __restrict__ q, int nq, const float* __restrict__ rows_img,
                                                         const _Float16* __restrict__ rows16, const _Float16* __restrict__ rows16l, const float* __restrict__ hd, long long n,
                                                         unsigned row_base, int rows_per_slice, const float* __restrict__ t0, int t0_stride, u64* __restrict__ parts,
                                                         const float* __restrict__ qc) {
    __shared__ u64 s_lists[4][64 * K2];
    __shared__ float s_aq[4][64], s_hq[4][64], s_t0[4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slice = blockIdx.x;
    const int q0 = (blockIdx.y * 4 + wave) * 64;                     // this wave's 64 queries
    if (q0 >= nq) return;                                            // no barrier in this kernel: a wave may leave
    const long long n32 = (n + 31) / 32 * 32;
    const long long r_lo = (long long)slice * rows_per_slice;        // multiple of 64
    long long r_hi = r_lo + rows_per_slice;
    if (r_hi > n32) r_hi = n32;
    u64* lists = s_lists[wave];
    float* aqs = s_aq[wave];
    float* hqs = s_hq[wave];
    float* t0s = s_t0[wave];
    const int li = lane & 15, lg = lane >> 4;

    // ---------------------------------------------------------------- lists start empty, thresholds from the sample pass
    // t0[q] = k2-th best exact distance among the shard's first rows (rf_l2_topk's sample pass, VALU scan): an upper bound of
    // the final k2-th distance (+inf when the sample holds fewer than k2 rows).
    for (int ql = lane; ql < 64; ql += 64) {
        const int qi = q0 + ql;
        float hq = 0.f, a = INFINITY;                                // a query that does not exist: nothing passes
        if (qi < nq) {
            const float* qp = q + (size_t)qi * RF_DIM;
            float nqn = 0.f, big = 0.f;
#pragma unroll
            for (int d = 0; d < RF_DIM; ++d) { nqn = fmaf(qp[d], qp[d], nqn); big = fmaxf(big, fabsf(qp[d])); }
            hq = big < 6.0e4f ? (1.f - RF_EPS_FILTER16) * 0.5f * nqn : -INFINITY;     // beyond the f16 range: everything is re-checked
            const float t = t0[(size_t)qi * t0_stride];
            a = hq - 0.5f * t;                                       // t = +inf (sample smaller than k2): -inf, everything passes
        }
        hqs[ql] = hq;
        aqs[ql] = a;
        t0s[ql] = qi < nq ? t0[(size_t)qi * t0_stride] : 0.f;
    }
    for (int i = lane; i < 64 * K2; i += 64) lists[i] = RF_KEY_NONE;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---------------------------------------------------------------- the filtered scan
    // (the fp32 queries are NOT held in registers: the exact re-check reads a query's summation chains from `qc` -- [query][chain g][16] = dims {4 j + g}, written by
    // k_query_chains -- and the 64 registers hold a third tile of rows instead: the scan waits for memory 76 % of its time with one tile requested ahead)
    rf_h8 bq[4][2], bql[4][2];                                       // the filter's B operands: k = 32 t + 8 lg + j of query q0 + nb*16 + li, as f16 pieces h, l
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int qi = q0 + nb * 16 + li;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float qv = qi < nq && hqs[nb * 16 + li] != -INFINITY ? q[(size_t)qi * RF_DIM + 32 * t + 8 * lg + j] : 0.f;   // out-of-range query: zeros, a_q = -inf
                const _Float16 qh = (_Float16)qv;
                bq[nb][t][j] = qh;
                bql[nb][t][j] = (_Float16)((qv - (float)qh) * 2048.f);
            }
    }
    float aq[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) aq[nb] = aqs[nb * 16 + li];

    auto load_tile = [&](long long row0, rf_h8 (&a16)[2][2], rf_h8 (&a16l)[2][2], f32x4 (&h)[2]) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const rf_h8* rp = reinterpret_cast<const rf_h8*>(rows16 + (size_t)(row0 + mb * 16 + li) * RF_DIM + 8 * lg);
            a16[mb][0] = rp[0];
            a16[mb][1] = rp[4];                                       // + 32 halves
            const rf_h8* rl = reinterpret_cast<const rf_h8*>(rows16l + (size_t)(row0 + mb * 16 + li) * RF_DIM + 8 * lg);
            a16l[mb][0] = rl[0];
            a16l[mb][1] = rl[4];
            const float4 t = *reinterpret_cast<const float4*>(hd + row0 + mb * 16 + 4 * lg);
            h[mb] = (f32x4){-t.x, -t.y, -t.z, -t.w};
        }
    };
    // the fp32 rows of a tile, fetched only when some pair passed the filter (the exact re-check reads them)
    auto load_exact = [&](long long row0, float (&a)[2][16]) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const float4* rp = reinterpret_cast<const float4*>(rows_img + (size_t)(row0 + mb * 16 + li) * RF_DIM + 16 * lg);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float4 t = rp[k]; a[mb][4 * k] = t.x; a[mb][4 * k + 1] = t.y; a[mb][4 * k + 2] = t.z; a[mb][4 * k + 3] = t.w; }
        }
    };

    auto scan_tile = [&](long long row0, const rf_h8 (&a16)[2][2], const rf_h8 (&a16l)[2][2], const f32x4 (&h)[2]) {
        // the tile in four quarters (m-block x pair of n-blocks): 2 x 2 accumulators live at a time -- the kernel holds the 64 queries three times over
        // (fp32 for the exact re-check, h and l pieces for the filter) and two tiles of rows
        unsigned m = 0u;                                              // bit (mb*4 + nb)*4 + i: D row 4*lg + i of m-block mb, query column li of n-block nb
        unsigned long long anyhit = 0ull;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int np = 0; np < 2; ++np) {
                f32x4 acc[2], acl[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) { acc[e] = h[mb]; acl[e] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[mb][t], bq[2 * np + e][t], acc[e], 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 2; ++e) acl[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[mb][t], bql[2 * np + e][t], acl[e], 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 2; ++e) acl[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16l[mb][t], bq[2 * np + e][t], acl[e], 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bool pass = fmaf(acl[e][i], 1.0f / 2048.f, acc[e][i]) >= aq[2 * np + e];
                        anyhit |= __ballot(pass);
                        m |= (pass ? 1u : 0u) << ((mb * 4 + 2 * np + e) * 4 + i);
                    }
            }
        if (anyhit == 0ull) return;
        // ---- some pair passed the filter: fetch the tile's fp32 rows, exact re-check from registers
        float a[2][16];
        load_exact(row0, a);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const unsigned m4 = (m >> ((mb * 4 + nb) * 4)) & 15u;
                const unsigned long long bal = __ballot(m4 != 0u);
                unsigned qcols = (unsigned)((bal | (bal >> 16) | (bal >> 32) | (bal >> 48)) & 0xffffull);      // query columns n_ with a passing pair in this m-block
                while (qcols) {
                    const int n_ = __ffs((int)qcols) - 1;
                    qcols &= qcols - 1;
                    const int ql = nb * 16 + n_;
                    int qi = q0 + ql;
                    qi = qi < nq ? qi : nq - 1;
                    // chain g of THE exact distance of query n_ against the m-block's 16 rows, ONCE: lane (r, g) runs chain g of row r
                    const float4* qp = reinterpret_cast<const float4*>(qc + (size_t)qi * RF_DIM + 16 * lg);
                    float P = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float4 qv = qp[k];
                        float t = qv.x - a[mb][4 * k]; P = fmaf(t, t, P);
                        t = qv.y - a[mb][4 * k + 1]; P = fmaf(t, t, P);
                        t = qv.z - a[mb][4 * k + 2]; P = fmaf(t, t, P);
                        t = qv.w - a[mb][4 * k + 3]; P = fmaf(t, t, P);
                    }
                    const unsigned Pb = __float_as_uint(P);            // readlane moves 32-bit patterns
#pragma unroll
                    for (int g_ = 0; g_ < 4; ++g_) {
                        unsigned mi = __builtin_amdgcn_readlane(m4, n_ + 16 * g_);
                        while (mi) {
                            const int i = __ffs((int)mi) - 1;
                            mi &= mi - 1;
                            const int r = 4 * g_ + i;                 // row of the m-block
                            const long long row = row0 + mb * 16 + r;
                            if (row >= n) continue;
                            const float c0 = __uint_as_float(__builtin_amdgcn_readlane(Pb, r)), c1 = __uint_as_float(__builtin_amdgcn_readlane(Pb, r + 16));
                            const float c2 = __uint_as_float(__builtin_amdgcn_readlane(Pb, r + 32)), c3 = __uint_as_float(__builtin_amdgcn_readlane(Pb, r + 48));
                            const float dist = (c0 + c2) + (c1 + c3);
                            const u64 key = make_key(dist, row_base + (unsigned)row);
                            u64 e = lane < K2 ? lists[ql * K2 + lane] : RF_KEY_NONE;
                            const unsigned wlo = __builtin_amdgcn_readlane((unsigned)(e & 0xffffffffu), K2 - 1);
                            const unsigned whi = __builtin_amdgcn_readlane((unsigned)(e >> 32), K2 - 1);
                            if (key < (((u64)whi << 32) | wlo) && dist <= t0s[ql]) {          // (<=: the sample's own k2-th row must get in)
                                list_insert<K2>(e, lane, key);
                                if (lane < K2) lists[ql * K2 + lane] = e;
                                if (lane == K2 - 1) {                    // T = min(sample bound, the list's k2-th distance once it is full)
                                    const float tl = e == RF_KEY_NONE ? INFINITY : __uint_as_float((unsigned)(e >> 32));
                                    aqs[ql] = hqs[ql] - 0.5f * fminf(tl, t0s[ql]);
                                }
                            }
                        }
                    }
                }
            }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) aq[nb] = aqs[nb * 16 + li];
    };

    // three tiles of rows in registers: two requested ahead
    rf_h8 a0[2][2], a1[2][2], a2[2][2], l0[2][2], l1[2][2], l2[2][2];
    f32x4 h0[2], h1[2], h2[2];
    long long row0 = r_lo;
    if (row0 < r_hi) load_tile(row0, a0, l0, h0);
    if (row0 + 32 < r_hi) load_tile(row0 + 32, a1, l1, h1);
    while (row0 < r_hi) {
        if (row0 + 64 < r_hi) load_tile(row0 + 64, a2, l2, h2);
        scan_tile(row0, a0, l0, h0);
        row0 += 32;
        if (row0 >= r_hi) break;
        if (row0 + 64 < r_hi) load_tile(row0 + 64, a0, l0, h0);
        scan_tile(row0, a1, l1, h1);
        row0 += 32;
        if (row0 >= r_hi) break;
        if (row0 + 64 < r_hi) load_tile(row0 + 64, a1, l1, h1);
        scan_tile(row0, a2, l2, h2);
        row0 += 32;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // publish: parts[slice][q][K2]
    for (int ql = 0; ql < 64; ++ql) {
        const int qi = q0 + ql;
        if (qi < nq && lane < K2) parts[((size_t)slice * nq + qi) * K2 + lane] = lists[ql * K2 + lane];
    }
}

// The SEED of the filtered scan (round 6; it replaced an exact VALU / nested filtered scan of the shard's first n/8 rows: 0.14 ms of the 0.41 ms search at
// 2048 x 50 k, a third of it at 1 M rows).  The filtered scan needs, per query, an upper bound T0 of the final k2-th distance.  ANY k2 distinct rows give one
// (the largest of their exact distances), and the bound is as tight as the rows are good -- so the rows are chosen by the split-operand filter alone, and only the
// chosen ones are evaluated exactly:
//   * the same tile walk as k_l2_topk_mfma16 over the seed rows (a wave = 64 queries x a slice of rows; s = q.x - hd from 6 MFMAs per 16 x 16 tile), but a value
//     is only compared with the running best of its ACCUMULATOR SLOT: lane (li, lg), slot (n-block, i) sees rows 16 u + 4 lg + i (u = 0, 1, ..) of the slice
//     for query 16 nb + li -- 16 disjoint row classes per (slice, query), a best s and its u each (4 VALU per pair, no ballot, no list, no exact step);
//   * the two best slots of every (lane, n-block) are published as keys (descending s, local row): 8 candidates per (slice, query), from 8 different classes.
//     k_merge_wave takes the k2 best of the slices x 8, evaluates THE exact distance of those k2 rows, and writes T0 = the largest.
// How tight: the true k2 nearest seed rows are all found unless two of them share a class of one slice (same row index mod 16 within the slice) or three a lane:
// with S slices the k2-th best candidate has expected rank ~ k2 + k2^2 / (32 S) among the seed rows -- 8.03 for k2 = 8, S = 64 (the exact 8th, nearly) -- so the
// seed can cover HALF OR ALL of the shard at 0.4 matrix-pipe cycles per pair, and the filtered scan behind it re-checks ~k2 n / seed pairs per query instead of 64.
// Rows the filter cannot rank (hd16 = -inf: s = +inf) win their class and are then measured exactly like any other: the bound stays valid, only looser.
__device__ __forceinline__ unsigned rf_desc_bits(float s) {          // monotone: larger s -> smaller unsigned
    const unsigned u = __float_as_uint(-s);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

__global__ __launch_bounds__(256, 2) void k_l2_topk_seed16(const float* __restrict__ q, int nq, const _Float16* __restrict__ rows16, const _Float16* __restrict__ rows16l,
                                                         const float* __restrict__ hd, long long n, int rows_per_slice, u64* __restrict__ parts) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slice = blockIdx.x;
    const int q0 = (blockIdx.y * 4 + wave) * 64;
    if (q0 >= nq) return;
    const long long n32 = (n + 31) / 32 * 32;
    const long long r_lo = (long long)slice * rows_per_slice;        // multiple of 32
    long long r_hi = r_lo + rows_per_slice;
    if (r_hi > n32) r_hi = n32;
    const int li = lane & 15, lg = lane >> 4;

    rf_h8 bq[4][2], bql[4][2];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int qi = q0 + nb * 16 + li;
        float qv[2][8];
        float big = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                qv[t][j] = qi < nq ? q[(size_t)qi * RF_DIM + 32 * t + 8 * lg + j] : 0.f;
                big = fmaxf(big, fabsf(qv[t][j]));
            }
        big = fmaxf(big, __shfl_xor(big, 16, 64));                    // the query's four lanes hold its 64 dims between them
        big = fmaxf(big, __shfl_xor(big, 32, 64));
        const bool fits = big < 6.0e4f;                               // beyond the f16 range: zeros (the seed rows of such a query are arbitrary, the bound stays valid)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = fits ? qv[t][j] : 0.f;
                const _Float16 qh = (_Float16)v;
                bq[nb][t][j] = qh;
                bql[nb][t][j] = (_Float16)((v - (float)qh) * 2048.f);
            }
    }
    float best[4][4];                                                // slot (n-block, i): rows 16 u + 4 lg + i of the slice, u = 2 tile + m-block
    int bu[4][4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int i = 0; i < 4; ++i) { best[nb][i] = -INFINITY; bu[nb][i] = -1; }

    auto load_tile = [&](long long row0, rf_h8 (&a16)[2][2], rf_h8 (&a16l)[2][2], f32x4 (&h)[2]) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const rf_h8* rp = reinterpret_cast<const rf_h8*>(rows16 + (size_t)(row0 + mb * 16 + li) * RF_DIM + 8 * lg);
            a16[mb][0] = rp[0];
            a16[mb][1] = rp[4];
            const rf_h8* rl = reinterpret_cast<const rf_h8*>(rows16l + (size_t)(row0 + mb * 16 + li) * RF_DIM + 8 * lg);
            a16l[mb][0] = rl[0];
            a16l[mb][1] = rl[4];
            const float4 t = *reinterpret_cast<const float4*>(hd + row0 + mb * 16 + 4 * lg);
            h[mb] = (f32x4){-t.x, -t.y, -t.z, -t.w};
        }
    };
    auto scan_tile = [&](int tile, const rf_h8 (&a16)[2][2], const rf_h8 (&a16l)[2][2], const f32x4 (&h)[2]) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int np = 0; np < 2; ++np) {
                f32x4 acc[2], acl[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) { acc[e] = h[mb]; acl[e] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[mb][t], bq[2 * np + e][t], acc[e], 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 2; ++e) acl[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[mb][t], bql[2 * np + e][t], acl[e], 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 2; ++e) acl[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16l[mb][t], bq[2 * np + e][t], acl[e], 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float sv = fmaf(acl[e][i], 1.0f / 2048.f, acc[e][i]);
                        const bool better = sv > best[2 * np + e][i];
                        best[2 * np + e][i] = better ? sv : best[2 * np + e][i];
                        bu[2 * np + e][i] = better ? 2 * tile + mb : bu[2 * np + e][i];
                    }
            }
    };
    rf_h8 a0[2][2], a1[2][2], l0[2][2], l1[2][2];
    f32x4 h0[2], h1[2];
    long long row0 = r_lo;
    int tile = 0;
    if (row0 < r_hi) load_tile(row0, a0, l0, h0);
    while (row0 < r_hi) {
        if (row0 + 32 < r_hi) load_tile(row0 + 32, a1, l1, h1);
        scan_tile(tile, a0, l0, h0);
        row0 += 32; ++tile;
        if (row0 >= r_hi) break;
        if (row0 + 32 < r_hi) load_tile(row0 + 32, a0, l0, h0);
        scan_tile(tile, a1, l1, h1);
        row0 += 32; ++tile;
    }
    // publish the two best slots of every (lane, n-block): parts[slice][q][8], class = 2 lg + rank
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int qi = q0 + nb * 16 + li;
        float s1 = -INFINITY, s2 = -INFINITY;
        int r1 = -1, r2 = -1;                                        // rows relative to r_lo
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float sv = best[nb][i];
            const int row = bu[nb][i] * 16 + 4 * lg + i;
            const bool have = bu[nb][i] >= 0;
            const bool first = have && sv > s1, second = have && !first && sv > s2;
            s2 = first ? s1 : (second ? sv : s2);
            r2 = first ? r1 : (second ? row : r2);
            s1 = first ? sv : s1;
            r1 = first ? row : r1;
        }
        if (qi >= nq) continue;
        u64* out = parts + ((size_t)slice * nq + qi) * 8 + 2 * lg;
        out[0] = r1 >= 0 && r_lo + r1 < n ? (((u64)rf_desc_bits(s1) << 32) | (unsigned)(r_lo + r1)) : RF_KEY_NONE;
        out[1] = r2 >= 0 && r_lo + r2 < n ? (((u64)rf_desc_bits(s2) << 32) | (unsigned)(r_lo + r2)) : RF_KEY_NONE;
    }
}

// Merge of candidate lists: the k2 smallest keys per query (as (dist, idx) pairs and / or packed keys).  Rounds 1-5 sorted up to 4096 keys per query in LDS
// (bitonic, 55 barrier stages for 1024 keys: 58 us for 2048 queries x 61 lists).
// One WAVE per query holds the candidates in registers (R per lane) and takes the smallest key k2 times (a lane-local minimum,
// a wave minimum, one copy of the taken key struck out) -- no LDS, no barrier; 512 candidates x 2048 queries: 58 -> ~6 us.
// seed_q != nullptr: the candidates are k_l2_topk_seed16's (descending s, LOCAL row) keys; the k2 best rows are measured with THE exact distance (lane k takes
// row k: chains c over dims 4 m + c from the chain-ordered images `seed_rows` / `seed_q`) and seed_t0[query] = the largest of them (+inf with fewer than k2 rows).
__device__ __forceinline__ u64 wave_min_u64(u64 v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u64 o = __shfl_xor(v, d, 64);
        v = o < v ? o : v;
    }
    return v;
}

template <int R>
__global__ __launch_bounds__(256) void k_merge_wave(const u64* __restrict__ parts, int nparts, int nq, int width, int k2, float* __restrict__ out_dist,
                                                    long long* __restrict__ out_idx, u64* __restrict__ out_keys, const float* __restrict__ seed_q,
                                                    const float* __restrict__ seed_rows, float* __restrict__ seed_t0) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const int total = nparts * width;
    u64 key[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int c = r * 64 + lane;
        key[r] = RF_KEY_NONE;
        if (c < total) { const int part = c / width, j = c - part * width; key[r] = parts[((size_t)part * nq + qi) * width + j]; }
    }
    u64 mine = RF_KEY_NONE;
    for (int k = 0; k < k2; ++k) {
        u64 lm = key[0];
#pragma unroll
        for (int r = 1; r < R; ++r) lm = key[r] < lm ? key[r] : lm;
        const u64 wm = wave_min_u64(lm);
        if (lane == k) mine = wm;
        if (wm == RF_KEY_NONE) break;                                // wave-uniform: nothing left
        bool has = false;                                            // strike out ONE copy (merged lists may repeat a key; the sort this replaces kept both)
#pragma unroll
        for (int r = 0; r < R; ++r) has = has || key[r] == wm;
        const unsigned long long holders = __ballot(has);
        if (lane == __ffsll((long long)holders) - 1) {
            bool done = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool hit = !done && key[r] == wm;
                key[r] = hit ? RF_KEY_NONE : key[r];
                done = done || hit;
            }
        }
    }
    if (seed_t0) {
        float dist = -INFINITY;
        if (lane < k2) {
            dist = INFINITY;
            if (mine != RF_KEY_NONE) {
                const float4* xp = reinterpret_cast<const float4*>(seed_rows + (size_t)(unsigned)(mine & 0xffffffffu) * RF_DIM);
                const float4* qp = reinterpret_cast<const float4*>(seed_q + (size_t)qi * RF_DIM);
                float c[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float P = 0.f;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const float4 qv = qp[4 * g + kk], xv = xp[4 * g + kk];
                        float t = qv.x - xv.x; P = fmaf(t, t, P);
                        t = qv.y - xv.y; P = fmaf(t, t, P);
                        t = qv.z - xv.z; P = fmaf(t, t, P);
                        t = qv.w - xv.w; P = fmaf(t, t, P);
                    }
                    c[g] = P;
                }
                dist = (c[0] + c[2]) + (c[1] + c[3]);
            }
        }
        dist = wave_max(dist);
        if (lane == 0) seed_t0[qi] = dist;
        return;
    }
    if (lane < k2) {
        const bool none = mine == RF_KEY_NONE;
        if (out_keys) out_keys[(size_t)qi * k2 + lane] = mine;
        if (out_dist) out_dist[(size_t)qi * k2 + lane] = none ? INFINITY : __uint_as_float((unsigned)(mine >> 32));
        if (out_idx) out_idx[(size_t)qi * k2 + lane] = none ? -1ll : (long long)(unsigned)(mine & 0xffffffffu);
    }
}

extern "C" size_t rf_l2_topk_ws_bytes(int nq, int64_t n, int k2) {
    (void)n;
    const int k2p = k2 <= 8 ? 8 : 16;
    return (size_t)64 * (size_t)nq * k2p * sizeof(u64)            // one K2-wide list per (slice <= 64, query)
           + (size_t)nq * k2p * (sizeof(float) + sizeof(int64_t))    // + the sample pass's result (MFMA-filtered scan)
           + (size_t)nq * 64 * sizeof(float);                        // + the queries in chain order (k_query_chains)
}

template <int R>
static void launch_merge_wave(const u64* parts, int nparts, int nq, int width, int k2, float* out_dist, long long* out_idx, u64* out_keys, const float* seed_q,
                              const float* seed_rows, float* seed_t0, hipStream_t s) {
    hipLaunchKernelGGL(k_merge_wave<R>, dim3((nq + 3) / 4), dim3(256), 0, s, parts, nparts, nq, width, k2, out_dist, out_idx, out_keys, seed_q, seed_rows, seed_t0);
}

static int launch_merge(const u64* parts, int nparts, int nq, int width, int k2, float* out_dist, int64_t* out_idx, u64* out_keys, hipStream_t s,
                        const char* who, const float* seed_q = nullptr, const float* seed_rows = nullptr, float* seed_t0 = nullptr) {
    const int total = nparts * width;
    RF_REQUIRE(total <= 4096, RF_E_UNSUPPORTED, "%s: %d candidates per query exceed the merge capacity 4096", who, total);
    RF_REQUIRE(k2 >= 1 && k2 <= 64, RF_E_UNSUPPORTED, "%s: k2 = %d outside 1..64", who, k2);
    long long* oi = (long long*)out_idx;
    if (total <= 128) launch_merge_wave<2>(parts, nparts, nq, width, k2, out_dist, oi, out_keys, seed_q, seed_rows, seed_t0, s);
    else if (total <= 256) launch_merge_wave<4>(parts, nparts, nq, width, k2, out_dist, oi, out_keys, seed_q, seed_rows, seed_t0, s);
    else if (total <= 512) launch_merge_wave<8>(parts, nparts, nq, width, k2, out_dist, oi, out_keys, seed_q, seed_rows, seed_t0, s);
    else if (total <= 1024) launch_merge_wave<16>(parts, nparts, nq, width, k2, out_dist, oi, out_keys, seed_q, seed_rows, seed_t0, s);
    else if (total <= 2048) launch_merge_wave<32>(parts, nparts, nq, width, k2, out_dist, oi, out_keys, seed_q, seed_rows, seed_t0, s);
    else launch_merge_wave<64>(parts, nparts, nq, width, k2, out_dist, oi, out_keys, seed_q, seed_rows, seed_t0, s);
    RF_CHECK_LAUNCH(who);
    return RF_OK;
}

// Which scan (measured on MI355X, tools/topk_bench.py; k2 = 8): 2048 queries -- 50 k rows: VALU 0.47 ms / MFMA 0.59 ms, 125 k: 0.92 / 0.79,
// 1 M: 6.3 / 2.7;  16384 queries (8 ranks' queries against one shard) -- 50 k: 2.8 / 2.0, 125 k: 6.4 / 3.4.
// (round 5, after the sample pass went from n/64 to n/8 rows and became recursive: 2048 queries -- 12.5 k rows: VALU 0.19 ms / f16 filter 0.26, 50 k: 0.46 / 0.31,
// 1 M: 6.2 / 1.10; 1024 queries x 50 k: 0.30 / 0.24; 16384 queries x 125 k (one of 8 shards, all ranks' queries): 6.0 / 2.0.)  The filtered scan pays from ~4e7
// (query, row) pairs on, whichever way they split.
// Short shards under many queries stay on the VALU scan (one rank of 8 on the 50 k-row database: 16384 queries x 6250 rows 0.46 / 0.60 ms; x 12.5 k: 0.82 / 0.72:
// the filtered scan pays per query for its lists and re-checks): tools/topk_shard_shapes.py.
// (round 6, split-operand filter, tools/topk_shard_shapes.py: 2048 x 50 k 0.37 / VALU 0.45 ms; 4096 x 25 k 0.41 / 0.43; 8192 x 12.5 k 0.55 / 0.44; 16384 x 125 k 2.9 / 6.1)
static bool use_mfma_scan(int nq, int64_t n) { return n >= 20000 && (double)nq * (double)n >= 4.0e7; }

// workspace: [per-slice lists: 64 x nq x k2p keys][sample pass: nq x k2p (dist f32, idx i64)]
// n_layout: the row count the packed image was built for (the offsets of its views depend on it); n <= n_layout: the rows scanned (the first n of the shard)
static int topk_impl(const float* q, int nq, int dim, const float* db_packed, int64_t n, int64_t n_layout, int64_t row_base, int k2, int algo,
                     float* out_dist, int64_t* out_idx, u64* out_keys, void* ws, size_t ws_bytes, void* stream) {
    RF_REQUIRE(q && db_packed && (out_keys || (out_dist && out_idx)) && ws && nq > 0 && n > 0, RF_E_INVALID, "rf_l2_topk: bad arguments");
    RF_REQUIRE(dim == RF_DIM, RF_E_UNSUPPORTED, "rf_l2_topk: embedding dim %d (only 64, the latent_dim of every shipped config)", dim);
    RF_REQUIRE(k2 >= 1 && k2 <= 16, RF_E_UNSUPPORTED, "rf_l2_topk: k2=%d outside 1..16", k2);
    RF_REQUIRE(algo >= 0 && algo <= 3, RF_E_INVALID, "rf_l2_topk: algo %d (0 auto, 1 VALU scan, 2 fp32-MFMA-filtered scan, 3 f16-MFMA-filtered scan)", algo);
    RF_REQUIRE(row_base >= 0 && row_base + n <= 0xFFFFFFFELL, RF_E_UNSUPPORTED, "rf_l2_topk: global row ids must fit 32 bits");
    RF_REQUIRE(ws_bytes >= rf_l2_topk_ws_bytes(nq, n, k2), RF_E_WORKSPACE, "rf_l2_topk: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const long long nblk = (n + 63) / 64;
    const int k2p = k2 <= 8 ? 8 : 16;
    u64* parts = (u64*)ws;
    int slices;
    const bool mfma = algo == 2 || algo == 3 || (algo == 0 && use_mfma_scan(nq, n));
    const bool f16_filter = algo != 2;
    if (mfma) {
        // one wave per (slice, 64 queries); two waves per SIMD on 256 CUs = 2048 waves; at least 128 rows per list, at most 64 lists
        const int qgroups = (nq + 63) / 64, qtiles = (qgroups + 3) / 4;
        long long sl = (2048 + qgroups - 1) / qgroups;
        if (sl > 64) sl = 64;
        if (sl > nblk / 2) sl = nblk / 2;
        if (sl < 1) sl = 1;
        const long long bps = (nblk + sl - 1) / sl;
        slices = (int)((nblk + bps - 1) / bps);
        const float* rows_img = db_packed + rf_blocked_floats(n_layout);
        const float* hd = rows_img + (size_t)rf_rows32(n_layout) * RF_DIM;
        // The threshold pass.  fp32-filtered scan (algo 2, kept for comparison): exact top-k2p of the first n/8 rows by this function itself (VALU scan, or
        // nested filtered scans for big samples); its k2-th distance bounds the final one.  f16-filtered scan: k_l2_topk_seed16 + k_merge_wave (see there) over the
        // first n / RF_TOPK_SEED_DIV rows.
#ifndef RF_TOPK_SAMPLE_DIV
#define RF_TOPK_SAMPLE_DIV 8
#endif
#ifndef RF_TOPK_SEED_DIV
#define RF_TOPK_SEED_DIV 2
#endif
        float* t_dist = reinterpret_cast<float*>(parts + (size_t)64 * nq * k2p);
        int64_t* t_idx = reinterpret_cast<int64_t*>(t_dist + (size_t)nq * k2p) ;
        float* qc = reinterpret_cast<float*>(t_idx + (size_t)nq * k2p);
        const _Float16* rows16 = reinterpret_cast<const _Float16*>(hd + rf_rows32(n_layout));
        const float* hd16 = reinterpret_cast<const float*>(rows16 + (size_t)rf_rows32(n_layout) * RF_DIM);
        const _Float16* rows16l = reinterpret_cast<const _Float16*>(hd16 + rf_rows32(n_layout));
        const float* t0;
        int t0_stride;
        if (f16_filter) {
            hipLaunchKernelGGL(k_query_chains, dim3((unsigned)(((size_t)nq * RF_DIM + 255) / 256)), dim3(256), 0, s, q, nq, qc);
            long long seed = n / (n <= 131072 ? RF_TOPK_SEED_DIV : 4 * RF_TOPK_SEED_DIV);     // measured (tools/topk_bench.py): 50 k rows n/2 0.19 ms, n/8 0.22; 1 M rows n/8 1.28, n/2 1.53
            if (seed < 2048) seed = 2048;
            seed = (seed + 63) / 64 * 64;
            if (seed > n) seed = n;
            // 8 candidates per (slice, query) in the lists' workspace (64 k2p keys per query): at most 64 slices, at least 8 tiles each, ~2048 waves if possible
            const long long tiles = (seed + 31) / 32;
            long long ssl = (2048 + qgroups - 1) / qgroups;
            if (ssl > 64) ssl = 64;
            if (ssl > tiles / 8) ssl = tiles / 8;
            if (ssl < 1) ssl = 1;
            const long long tps = (tiles + ssl - 1) / ssl;
            const int sslices = (int)((tiles + tps - 1) / tps);
            hipLaunchKernelGGL(k_l2_topk_seed16, dim3(sslices, qtiles), dim3(256), 0, s, q, nq, rows16, rows16l, hd16, (long long)seed, (int)(tps * 32), parts);
            RF_CHECK_LAUNCH("rf_l2_topk(seed)");
            int rc = launch_merge(parts, sslices, nq, 8, k2, nullptr, nullptr, nullptr, s, "rf_l2_topk(seed merge)", qc, rows_img, t_dist);
            if (rc != RF_OK) return rc;
            t0 = t_dist;
            t0_stride = 1;
        } else {
            long long sample = n / RF_TOPK_SAMPLE_DIV;
            if (sample < 1024) sample = 1024;
            sample = (sample + 63) / 64 * 64;
            if (sample > n) sample = n;
            const int sample_algo = use_mfma_scan(nq, sample) ? 2 : 1;
            int rc = topk_impl(q, nq, dim, db_packed, sample, n_layout, row_base, k2p, sample_algo, t_dist, t_idx, nullptr, ws, ws_bytes, stream);
            if (rc != RF_OK) return rc;
            t0 = t_dist + (k2 - 1);                                   // the k2-th best of query qi: t0[qi * k2p]
            t0_stride = k2p;
        }
        if (f16_filter) {
            if (k2p == 8) hipLaunchKernelGGL(k_l2_topk_mfma16<8>, dim3(slices, qtiles), dim3(256), 0, s, q, nq, rows_img, rows16, rows16l, hd16, (long long)n, (unsigned)row_base, (int)(bps * 64), t0, t0_stride, parts, qc);
            else hipLaunchKernelGGL(k_l2_topk_mfma16<16>, dim3(slices, qtiles), dim3(256), 0, s, q, nq, rows_img, rows16, rows16l, hd16, (long long)n, (unsigned)row_base, (int)(bps * 64), t0, t0_stride, parts, qc);
        } else if (k2p == 8) hipLaunchKernelGGL(k_l2_topk_mfma<8>, dim3(slices, qtiles), dim3(256), 0, s, q, nq, rows_img, hd, (long long)n, (unsigned)row_base, (int)(bps * 64), t0, t0_stride, parts);
        else hipLaunchKernelGGL(k_l2_topk_mfma<16>, dim3(slices, qtiles), dim3(256), 0, s, q, nq, rows_img, hd, (long long)n, (unsigned)row_base, (int)(bps * 64), t0, t0_stride, parts);
        RF_CHECK_LAUNCH("rf_l2_topk(mfma scan)");
    } else {
        const int qtiles = (nq + RF_TQ - 1) / RF_TQ;
        // enough workgroups to fill 256 CUs a few times over, at most 64 slices (= lists per query to merge), at least 4 blocks per list
        long long sl = (1024 + qtiles - 1) / qtiles;
        if (sl > 64) sl = 64;
        if (sl > (nblk + 3) / 4) sl = (nblk + 3) / 4;
        if (sl < 1) sl = 1;
        const int bps = (int)((nblk + sl - 1) / sl);
        slices = (int)((nblk + bps - 1) / bps);
        if (k2p == 8) hipLaunchKernelGGL(k_l2_topk<8>, dim3(slices, qtiles), dim3(256), 0, s, q, nq, db_packed, (long long)n, (unsigned)row_base, bps, parts);
        else hipLaunchKernelGGL(k_l2_topk<16>, dim3(slices, qtiles), dim3(256), 0, s, q, nq, db_packed, (long long)n, (unsigned)row_base, bps, parts);
        RF_CHECK_LAUNCH("rf_l2_topk(scan)");
    }
    // merge the per-slice lists (each k2p wide) and emit the first k2
    return launch_merge(parts, slices, nq, k2p, k2, out_dist, out_idx, out_keys, s, "rf_l2_topk(merge)");
}

extern "C" int rf_l2_topk(const float* q, int nq, int dim, const float* db_packed, int64_t n, int64_t row_base, int k2, int algo,
                          float* out_dist, int64_t* out_idx, void* ws, size_t ws_bytes, void* stream) {
    return topk_impl(q, nq, dim, db_packed, n, n, row_base, k2, algo, out_dist, out_idx, nullptr, ws, ws_bytes, stream);
}

extern "C" int rf_l2_topk_keys(const float* q, int nq, int dim, const float* db_packed, int64_t n, int64_t row_base, int k2, int algo,
                               uint64_t* out_keys, void* ws, size_t ws_bytes, void* stream) {
    return topk_impl(q, nq, dim, db_packed, n, n, row_base, k2, algo, nullptr, nullptr, (u64*)out_keys, ws, ws_bytes, stream);
}

extern "C" int rf_topk_merge_keys(const uint64_t* in_keys, int parts, int nq, int k2, float* out_dist, int64_t* out_idx, void* stream) {
    RF_REQUIRE(in_keys && out_dist && out_idx && parts > 0 && nq > 0 && k2 > 0, RF_E_INVALID, "rf_topk_merge_keys: bad arguments");
    return launch_merge((const u64*)in_keys, parts, nq, k2, k2, out_dist, out_idx, nullptr, (hipStream_t)stream, "rf_topk_merge_keys");
}

// merge from (dist, idx) arrays: the all-gathered per-shard results
template <int CAP>
__global__ __launch_bounds__(256) void k_merge_pairs(const float* __restrict__ in_dist, const long long* __restrict__ in_idx, int nparts, int nq,
                                                     int k2, float* __restrict__ out_dist, long long* __restrict__ out_idx) {
    __shared__ u64 keys[CAP];
    const int qi = blockIdx.x, tid = threadIdx.x;
    const int total = nparts * k2;
    for (int i = tid; i < CAP; i += 256) {
        u64 v = RF_KEY_NONE;
        if (i < total) {
            const int part = i / k2, j = i % k2;
            const size_t o = ((size_t)part * nq + qi) * k2 + j;
            const long long id = in_idx[o];
            if (id >= 0) v = make_key(in_dist[o], (unsigned)id);
        }
        keys[i] = v;
    }
    __syncthreads();
    for (int size = 2; size <= CAP; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < CAP / 2; i += 256) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool asc = (lo & size) == 0;
                const u64 a = keys[lo], b = keys[hi];
                if ((a > b) == asc) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
    if (tid < k2) {
        const u64 key = keys[tid];
        const bool none = key == RF_KEY_NONE;
        out_dist[(size_t)qi * k2 + tid] = none ? INFINITY : __uint_as_float((unsigned)(key >> 32));
        out_idx[(size_t)qi * k2 + tid] = none ? -1ll : (long long)(unsigned)(key & 0xffffffffu);
    }
}

extern "C" int rf_topk_merge(const float* in_dist, const int64_t* in_idx, int parts, int nq, int k2,
                             float* out_dist, int64_t* out_idx, void* stream) {
    RF_REQUIRE(in_dist && in_idx && out_dist && out_idx && parts > 0 && nq > 0 && k2 > 0, RF_E_INVALID, "rf_topk_merge: bad arguments");
    const int total = parts * k2;
    RF_REQUIRE(total <= 4096, RF_E_UNSUPPORTED, "rf_topk_merge: %d candidates per query exceed 4096", total);
    hipStream_t s = (hipStream_t)stream;
    const long long* idx = (const long long*)in_idx;
    long long* oidx = (long long*)out_idx;
    if (total <= 256) hipLaunchKernelGGL(k_merge_pairs<256>, dim3(nq), dim3(256), 0, s, in_dist, idx, parts, nq, k2, out_dist, oidx);
    else if (total <= 1024) hipLaunchKernelGGL(k_merge_pairs<1024>, dim3(nq), dim3(256), 0, s, in_dist, idx, parts, nq, k2, out_dist, oidx);
    else hipLaunchKernelGGL(k_merge_pairs<4096>, dim3(nq), dim3(256), 0, s, in_dist, idx, parts, nq, k2, out_dist, oidx);
    RF_CHECK_LAUNCH("rf_topk_merge");
    return RF_OK;
}

// ---------------------------------------------------------------------------------------- same-scene demotion
__global__ __launch_bounds__(256) void k_demote(const float* __restrict__ dist, const long long* __restrict__ idx, int nq, int k2,
                                                const int* __restrict__ db_meta, const int* __restrict__ query_scene,
                                                const unsigned char* __restrict__ query_keep, int K,
                                                int* __restrict__ out_meta, float* __restrict__ out_dist, long long* __restrict__ out_idx) {
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    if (query_keep && !query_keep[qi]) {
        // a patch the query-side occupancy filter dropped is never looked up: its K slots are the "no neighbour" entry, which the
        // patch gather turns into the truncation fill (util/retrieval.py:148,151)
        for (int w = 0; w < K; ++w) {
            int* m = out_meta + ((size_t)qi * K + w) * 7;
            m[0] = -1; m[1] = 0; m[2] = 16; m[3] = 0; m[4] = 16; m[5] = 0; m[6] = 16;
            out_dist[(size_t)qi * K + w] = INFINITY;
            out_idx[(size_t)qi * K + w] = -1;
        }
        return;
    }
    const int qs = query_scene ? query_scene[qi] : -1;
    int written = 0;
    // pass 0: neighbours NOT from the query's scene, in order; pass 1: the demoted ones, in order
    for (int pass = 0; pass < 2 && written < K; ++pass) {
        for (int j = 0; j < k2 && written < K; ++j) {
            const long long id = idx[(size_t)qi * k2 + j];
            const int scene = id >= 0 ? db_meta[(size_t)id * 7] : -1;
            const bool same = qs >= 0 && id >= 0 && scene == qs;
            if ((pass == 0) == same) continue;
            int* m = out_meta + ((size_t)qi * K + written) * 7;
            if (id >= 0) {
#pragma unroll
                for (int t = 0; t < 7; ++t) m[t] = db_meta[(size_t)id * 7 + t];
            } else {
                m[0] = -1; m[1] = 0; m[2] = 16; m[3] = 0; m[4] = 16; m[5] = 0; m[6] = 16;
            }
            out_dist[(size_t)qi * K + written] = dist[(size_t)qi * k2 + j];
            out_idx[(size_t)qi * K + written] = id;
            ++written;
        }
    }
}

extern "C" int rf_demote_same_scene(const float* dist, const int64_t* idx, int nq, int k2, const int32_t* db_meta,
                                    const int32_t* query_scene, const uint8_t* query_keep, int K, int32_t* out_meta, float* out_dist,
                                    int64_t* out_idx, void* stream) {
    RF_REQUIRE(dist && idx && db_meta && out_meta && out_dist && out_idx && nq > 0 && k2 > 0 && K > 0 && K <= k2, RF_E_INVALID,
               "rf_demote_same_scene: bad arguments (k2=%d K=%d)", k2, K);
    hipLaunchKernelGGL(k_demote, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)stream, dist, (const long long*)idx, nq, k2, db_meta,
                       query_scene, query_keep, K, out_meta, out_dist, (long long*)out_idx);
    RF_CHECK_LAUNCH("rf_demote_same_scene");
    return RF_OK;
}

// ------------------------------------------------------------------------------------------------ patch gather
// one workgroup per (chunk, k, slot): copies a 16^3 box of a DB scene.  T = float: the fp32 voxel store; T = _Float16: the store in the reference's own
// precision (scenes are float16 on disk and in memory, dataset/scene.py:61,71 -- widening is exact, so the two stores gather the same bits)
template <typename T>
__global__ __launch_bounds__(256) void k_gather_patches(const T* __restrict__ vols, long long n_scenes, const int* __restrict__ meta, int K,
                                                        float trunc_fill, float ratio, float mean, float stddev, int layout,
                                                        float* __restrict__ out) {
    const int slot = blockIdx.x & 63, k = (blockIdx.x >> 6) % K, chunk = (blockIdx.x >> 6) / K;
    const int* m = meta + (((size_t)chunk * 64 + slot) * K + k) * 7;
    const int scene = m[0], x0 = m[1], y0 = m[3], z0 = m[5];
    const bool have = scene >= 0 && scene < n_scenes;
    const T* src = vols + (size_t)(have ? scene : 0) * 64 * 64 * 64;
    float* dst;
    size_t dsx, dsy;                                            // destination strides of the two slow dims
    if (layout == 1) {
        dst = out + (((size_t)chunk * K + k) * 64 + slot) * 4096;
        dsx = 256; dsy = 16;
    } else {
        const int xx = (slot >> 4) * 16, yy = ((slot >> 2) & 3) * 16, zz = (slot & 3) * 16;
        dst = out + ((size_t)chunk * K + k) * 262144 + ((size_t)xx * 64 + yy) * 64 + zz;
        dsx = 4096; dsy = 64;
    }
    if ((z0 & 3) == 0 && (((size_t)vols | (size_t)out) & 15) == 0) {
        // four voxels of a row per thread (the patch rows are 16 voxels: 8- / 16-byte requests, 16-byte stores -- a quarter of the memory instructions; the usual case:
        // database patches lie on the 16-voxel grid of their scenes)
        for (int i = threadIdx.x; i < 1024; i += 256) {
            const int dz = (i & 3) * 4, dy = (i >> 2) & 15, dx = i >> 6;
            float v[4] = {trunc_fill, trunc_fill, trunc_fill, trunc_fill};
            if (have) {
                const T* p = src + ((size_t)(x0 + dx) * 64 + (y0 + dy)) * 64 + (z0 + dz);
                if constexpr (sizeof(T) == 4) {
                    const float4 q = *reinterpret_cast<const float4*>(p);
                    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                } else {
                    typedef _Float16 gp_h4 __attribute__((ext_vector_type(4)));
                    const gp_h4 q = *reinterpret_cast<const gp_h4*>(p);
                    v[0] = (float)q[0]; v[1] = (float)q[1]; v[2] = (float)q[2]; v[3] = (float)q[3];
                }
            }
            float4 o;
            o.x = __fdiv_rn(__fsub_rn(__fmul_rn(v[0], ratio), mean), stddev); o.y = __fdiv_rn(__fsub_rn(__fmul_rn(v[1], ratio), mean), stddev);
            o.z = __fdiv_rn(__fsub_rn(__fmul_rn(v[2], ratio), mean), stddev); o.w = __fdiv_rn(__fsub_rn(__fmul_rn(v[3], ratio), mean), stddev);
            *reinterpret_cast<float4*>(dst + dx * dsx + dy * dsy + dz) = o;
        }
        return;
    }
    for (int i = threadIdx.x; i < 4096; i += 256) {
        const int dz = i & 15, dy = (i >> 4) & 15, dx = i >> 8;
        float v = trunc_fill;
        if (have) v = (float)src[((size_t)(x0 + dx) * 64 + (y0 + dy)) * 64 + (z0 + dz)];
        v = __fmul_rn(v, ratio);
        dst[dx * dsx + dy * dsy + dz] = __fdiv_rn(__fsub_rn(v, mean), stddev);
    }
}

template <typename T>
static int gather_patches(const T* db_volumes, int64_t n_scenes, const int32_t* meta, int chunks, int K, float trunc_fill, float trunc_ratio, float mean,
                          float stddev, int layout, float* out, void* stream, const char* who) {
    RF_REQUIRE(db_volumes && meta && out && chunks > 0 && K > 0 && n_scenes > 0, RF_E_INVALID, "%s: bad arguments", who);
    RF_REQUIRE(layout == 0 || layout == 1, RF_E_INVALID, "%s: layout %d", who, layout);
    hipLaunchKernelGGL(k_gather_patches<T>, dim3((unsigned)chunks * K * 64), dim3(256), 0, (hipStream_t)stream, db_volumes, (long long)n_scenes, meta,
                       K, trunc_fill, trunc_ratio, mean, stddev, layout, out);
    RF_CHECK_LAUNCH(who);
    return RF_OK;
}

extern "C" int rf_gather_patches(const float* db_volumes, int64_t n_scenes, const int32_t* meta, int chunks, int K,
                                 float trunc_fill, float trunc_ratio, float mean, float stddev, int layout,
                                 float* out, void* stream) {
    return gather_patches<float>(db_volumes, n_scenes, meta, chunks, K, trunc_fill, trunc_ratio, mean, stddev, layout, out, stream, "rf_gather_patches");
}

extern "C" int rf_gather_patches_f16(const void* db_volumes_f16, int64_t n_scenes, const int32_t* meta, int chunks, int K,
                                     float trunc_fill, float trunc_ratio, float mean, float stddev, int layout,
                                     float* out, void* stream) {
    return gather_patches<_Float16>(reinterpret_cast<const _Float16*>(db_volumes_f16), n_scenes, meta, chunks, K, trunc_fill, trunc_ratio, mean, stddev, layout, out,
                                    stream, "rf_gather_patches_f16");
}

// create_retrieval_from_mapping for patch grids WITH overlap (reference util/retrieval.py:145-164 with dataset.no_overlap False: patch stride < patch size).  The patches
// of a scene are visited in the order of patch_from_scene_lookup; patch p overwrites its box of retrieval k only while the MEAN of the distances stored in that box is
// above its own distance (:156) -- an order-dependent sequential reduction, so one workgroup per retrieval k walks the patches in order (an offline step of the
// reference: retrievals_to_disk('compose'); the hot path's chunks have non-overlapping patches and use k_gather_patches).  The box mean is accumulated in float64
// in a fixed order (thread-strided partial sums, then a tree over the 256 partials).  dist: [K][sx][sy][sz] workspace.
template <typename T>
__global__ __launch_bounds__(256) void k_compose_overlap(const T* __restrict__ vols, long long n_scenes, const float* __restrict__ mapping, const int* __restrict__ boxes,
                                                         int P, int K, int sx, int sy, int sz, float trunc_fill, float ratio, float* __restrict__ out,
                                                         float* dist) {
    __shared__ double part[256];
    const int k = blockIdx.x, tid = threadIdx.x;
    const size_t vol = (size_t)sx * sy * sz;
    float* o = out + (size_t)k * vol;
    volatile float* d = dist + (size_t)k * vol;                     // written and re-read by different threads of the workgroup: not through the L1
    for (size_t i = tid; i < vol; i += 256) { o[i] = trunc_fill; d[i] = 100.f; }
    __threadfence();
    __syncthreads();
    for (int p = 0; p < P; ++p) {
        const int* b = boxes + (size_t)p * 6;
        const float* m = mapping + ((size_t)p * K + k) * 8;
        const int x0 = b[0], ex = b[1] - b[0], y0 = b[2], ey = b[3] - b[2], z0 = b[4], ez = b[5] - b[4];
        const int n = ex * ey * ez;
        const float cur = m[7];
        double sum = 0.0;
        for (int i = tid; i < n; i += 256) {
            const int dz = i % ez, dy = (i / ez) % ey, dx = i / (ez * ey);
            sum += (double)d[((size_t)(x0 + dx) * sy + (y0 + dy)) * sz + (z0 + dz)];
        }
        part[tid] = sum;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) part[tid] += part[tid + s];
            __syncthreads();
        }
        const bool take = n > 0 && part[0] / (double)n > (double)cur;      // uniform
        __syncthreads();                                                    // everyone has read part[0]
        if (take) {
            const int scene = (int)m[0], X0 = (int)m[1], Y0 = (int)m[3], Z0 = (int)m[5];
            const bool have = scene >= 0 && scene < n_scenes;
            const T* src = vols + (size_t)(have ? scene : 0) * 64 * 64 * 64;
            for (int i = tid; i < n; i += 256) {
                const int dz = i % ez, dy = (i / ez) % ey, dx = i / (ez * ey);
                float v = trunc_fill;
                if (have) v = (float)src[((size_t)(X0 + dx) * 64 + (Y0 + dy)) * 64 + (Z0 + dz)];
                const size_t at = ((size_t)(x0 + dx) * sy + (y0 + dy)) * sz + (z0 + dz);
                o[at] = __fmul_rn(v, ratio);
                d[at] = cur;
            }
            __threadfence();
            __syncthreads();
        }
    }
}

extern "C" int rf_compose_overlap(const void* db_volumes, int half, int64_t n_scenes, const float* mapping, const int32_t* boxes, int P, int K, int sx, int sy, int sz,
                                  float trunc_fill, float trunc_ratio, float* out, float* dist_ws, void* stream) {
    RF_REQUIRE(db_volumes && mapping && boxes && out && dist_ws && n_scenes > 0 && P >= 0 && K > 0 && sx > 0 && sy > 0 && sz > 0, RF_E_INVALID,
               "rf_compose_overlap: bad arguments");
    if (half)
        hipLaunchKernelGGL(k_compose_overlap<_Float16>, dim3((unsigned)K), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const _Float16*>(db_volumes),
                           (long long)n_scenes, mapping, boxes, P, K, sx, sy, sz, trunc_fill, trunc_ratio, out, dist_ws);
    else
        hipLaunchKernelGGL(k_compose_overlap<float>, dim3((unsigned)K), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float*>(db_volumes),
                           (long long)n_scenes, mapping, boxes, P, K, sx, sy, sz, trunc_fill, trunc_ratio, out, dist_ws);
    RF_CHECK_LAUNCH("rf_compose_overlap");
    return RF_OK;
}

// ------------------------------------------------------------------------------------------------- row gather
// out[m][:] = src[idx[m]][:] for rows of `width` floats (width % 4 == 0): fetches cached per-database-patch retrieval
// features (query independent, see rfuse/database.py:build_feature_cache).  One workgroup per output row.
// idx < 0 ("no neighbour") selects the LAST source row: the database keeps its all-trunc sentinel patch there
// (util/retrieval.py:21-26,45), which is what the full path substitutes for a missing neighbour.
__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ src, long long n_src, const long long* __restrict__ idx,
                                                     int width4, float* __restrict__ out) {
    const long long r = idx[blockIdx.x];
    const float4* s4 = reinterpret_cast<const float4*>(src) + (size_t)(r < 0 || r >= n_src ? n_src - 1 : r) * width4;
    float4* o4 = reinterpret_cast<float4*>(out) + (size_t)blockIdx.x * width4;
    for (int i = threadIdx.x; i < width4; i += 256) o4[i] = s4[i];
}

extern "C" int rf_gather_rows(const float* src, int64_t n_src, const int64_t* idx, int64_t m, int width, float* out, void* stream) {
    RF_REQUIRE(src && idx && out && n_src > 0 && m > 0 && width > 0 && (width & 3) == 0, RF_E_INVALID, "rf_gather_rows: bad arguments");
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)m), dim3(256), 0, (hipStream_t)stream, src, (long long)n_src, (const long long*)idx, width / 4, out);
    RF_CHECK_LAUNCH("rf_gather_rows");
    return RF_OK;
}
