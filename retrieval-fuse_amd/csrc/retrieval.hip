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
__global__ __launch_bounds__(256) void k_query_windows(const float* __restrict__ raw, int b, int s, int ps, int ctx, float pad_value,
                                                       float mean, float stddev, float* __restrict__ out) {
    const int np = s / ps, w = ps + 2 * ctx;
    const size_t w3 = (size_t)w * w * w, total = (size_t)b * np * np * np * w3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int w2 = (int)(i % w), w1 = (int)((i / w) % w), w0 = (int)((i / ((size_t)w * w)) % w);
        const size_t win = i / w3;
        const int p2 = (int)(win % np), p1 = (int)((win / np) % np), p0 = (int)((win / ((size_t)np * np)) % np);
        const size_t bb = win / ((size_t)np * np * np);
        const int d0 = p0 * ps + w0 - ctx, d1 = p1 * ps + w1 - ctx, d2 = p2 * ps + w2 - ctx;
        float v = pad_value;
        if ((unsigned)d0 < (unsigned)s && (unsigned)d1 < (unsigned)s && (unsigned)d2 < (unsigned)s)
            v = raw[((bb * s + d0) * s + d1) * s + d2];
        out[i] = __fdiv_rn(__fsub_rn(v, mean), stddev);
    }
}

extern "C" int rf_query_windows(const float* raw, int b, int s, int ps, int ctx, float pad_value, float mean, float stddev,
                                float* out, void* stream) {
    RF_REQUIRE(raw && out && b > 0 && s > 0 && ps > 0 && ctx >= 0 && s % ps == 0, RF_E_INVALID, "rf_query_windows: bad arguments");
    const int np = s / ps, w = ps + 2 * ctx;
    const size_t total = (size_t)b * np * np * np * w * w * w;
    const size_t want = (total + 255) / 256;
    hipLaunchKernelGGL(k_query_windows, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, (hipStream_t)stream, raw, b, s, ps, ctx,
                       pad_value, mean, stddev, out);
    RF_CHECK_LAUNCH("rf_query_windows");
    return RF_OK;
}

// ------------------------------------------------------------------------------------------------- DB packing
__global__ __launch_bounds__(256) void k_db_pack(const float* __restrict__ emb, long long n, int dim, float* __restrict__ packed) {
    const long long nblk = (n + 63) / 64;
    const size_t total = (size_t)nblk * dim * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i % 64), d = (int)((i / 64) % dim);
        const long long row = (long long)(i / ((size_t)64 * dim)) * 64 + r;
        packed[i] = row < n ? emb[(size_t)row * dim + d] : 0.f;
    }
}

extern "C" size_t rf_db_packed_floats(int64_t n, int dim) { return (size_t)((n + 63) / 64) * dim * 64; }

extern "C" int rf_db_pack_embeddings(const float* emb, int64_t n, int dim, float* packed, void* stream) {
    RF_REQUIRE(emb && packed && n > 0 && dim > 0, RF_E_INVALID, "rf_db_pack_embeddings: bad arguments");
    const size_t want = (rf_db_packed_floats(n, dim) + 255) / 256;
    hipLaunchKernelGGL(k_db_pack, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, (hipStream_t)stream, emb, (long long)n, dim, packed);
    RF_CHECK_LAUNCH("rf_db_pack_embeddings");
    return RF_OK;
}

// --------------------------------------------------------------------------------------------------- the scan
#define RF_DIM 64
#define RF_TQ 64          // queries per workgroup tile

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

// One workgroup = (DB slice, tile of RF_TQ queries); its 4 waves SPLIT THE QUERIES (RF_QW each) and every wave walks all
// 64-row blocks of the slice, so there is exactly one candidate list per (slice, query).  A wave's lists live in
// registers (list j: entry i in lane i of e[j]).  The first block initialises a list with a wave-wide sort; afterwards a
// row enters only if it beats the list's current worst (ballot), which becomes rare quickly (~K2/b hits for block b).
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
        // this lane's DB row: 64 coalesced loads, one per dim (the 4 waves read the same block: L1/L2 hits)
        // kept as register PAIRS: the distance loop runs on the packed-fp32 VALU (v_pk_add_f32 / v_pk_fma_f32: two dims per
        // instruction), queries as scalar-register pairs
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
                rf_f32x2 acc = {0.f, 0.f};                           // even / odd dims: two independent fp32 FMA chains
#pragma unroll
                for (int d = 0; d < RF_DIM / 2; ++d) {
                    const rf_f32x2 qv = {qp[2 * d], qp[2 * d + 1]};
                    const rf_f32x2 t = qv - x[d];
                    acc = __builtin_elementwise_fma(t, t, acc);
                }
                const u64 key = valid ? make_key(acc[0] + acc[1], grow) : RF_KEY_NONE;
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

// bitonic sort of up to CAP keys per query in LDS; the first k2 are the answer
template <int CAP>
__global__ __launch_bounds__(256) void k_merge_keys(const u64* __restrict__ parts, int nparts, int nq, int width, int k2,
                                                    float* __restrict__ out_dist, long long* __restrict__ out_idx) {
    __shared__ u64 keys[CAP];
    const int qi = blockIdx.x, tid = threadIdx.x;
    const int total = nparts * width;
    for (int i = tid; i < CAP; i += 256) {
        u64 v = RF_KEY_NONE;
        if (i < total) { const int part = i / width, j = i % width; v = parts[((size_t)part * nq + qi) * width + j]; }
        keys[i] = v;
    }
    __syncthreads();
    for (int size = 2; size <= CAP; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < CAP / 2; i += 256) {
                const int lo = 2 * i - (i & (stride - 1));          // index with bit `stride` cleared
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

static int pick_slices(long long nblk, int qtiles) {
    // enough workgroups to fill 256 CUs a few times over, at most 64 slices (= lists per query to merge)
    long long s = (1024 + qtiles - 1) / qtiles;
    if (s > 64) s = 64;
    if (s > (nblk + 3) / 4) s = (nblk + 3) / 4;                     // at least 4 blocks per list
    if (s < 1) s = 1;
    return (int)s;
}

extern "C" size_t rf_l2_topk_ws_bytes(int nq, int64_t n, int k2) {
    (void)n;
    const int k2p = k2 <= 8 ? 8 : 16;
    return (size_t)64 * (size_t)nq * k2p * sizeof(u64);           // one K2-wide list per (slice <= 64, query)
}

static int launch_merge(const u64* parts, int nparts, int nq, int width, int k2, float* out_dist, int64_t* out_idx, hipStream_t s, const char* who) {
    const int total = nparts * width;
    RF_REQUIRE(total <= 4096, RF_E_UNSUPPORTED, "%s: %d candidates per query exceed the merge capacity 4096", who, total);
    if (total <= 256) hipLaunchKernelGGL(k_merge_keys<256>, dim3(nq), dim3(256), 0, s, parts, nparts, nq, width, k2, out_dist, (long long*)out_idx);
    else if (total <= 1024) hipLaunchKernelGGL(k_merge_keys<1024>, dim3(nq), dim3(256), 0, s, parts, nparts, nq, width, k2, out_dist, (long long*)out_idx);
    else hipLaunchKernelGGL(k_merge_keys<4096>, dim3(nq), dim3(256), 0, s, parts, nparts, nq, width, k2, out_dist, (long long*)out_idx);
    RF_CHECK_LAUNCH(who);
    return RF_OK;
}

extern "C" int rf_l2_topk(const float* q, int nq, int dim, const float* db_packed, int64_t n, int64_t row_base, int k2,
                          float* out_dist, int64_t* out_idx, void* ws, size_t ws_bytes, void* stream) {
    RF_REQUIRE(q && db_packed && out_dist && out_idx && ws && nq > 0 && n > 0, RF_E_INVALID, "rf_l2_topk: bad arguments");
    RF_REQUIRE(dim == RF_DIM, RF_E_UNSUPPORTED, "rf_l2_topk: embedding dim %d (only 64, the latent_dim of every shipped config)", dim);
    RF_REQUIRE(k2 >= 1 && k2 <= 16, RF_E_UNSUPPORTED, "rf_l2_topk: k2=%d outside 1..16", k2);
    RF_REQUIRE(row_base >= 0 && row_base + n <= 0xFFFFFFFELL, RF_E_UNSUPPORTED, "rf_l2_topk: global row ids must fit 32 bits");
    RF_REQUIRE(ws_bytes >= rf_l2_topk_ws_bytes(nq, n, k2), RF_E_WORKSPACE, "rf_l2_topk: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const long long nblk = (n + 63) / 64;
    const int qtiles = (nq + RF_TQ - 1) / RF_TQ;
    const int slices = pick_slices(nblk, qtiles);
    const int bps = (int)((nblk + slices - 1) / slices);
    const int k2p = k2 <= 8 ? 8 : 16;
    u64* parts = (u64*)ws;
    if (k2p == 8) hipLaunchKernelGGL(k_l2_topk<8>, dim3(slices, qtiles), dim3(256), 0, s, q, nq, db_packed, (long long)n, (unsigned)row_base, bps, parts);
    else hipLaunchKernelGGL(k_l2_topk<16>, dim3(slices, qtiles), dim3(256), 0, s, q, nq, db_packed, (long long)n, (unsigned)row_base, bps, parts);
    RF_CHECK_LAUNCH("rf_l2_topk(scan)");
    // merge the per-slice lists (each k2p wide) and emit the first k2
    return launch_merge(parts, slices, nq, k2p, k2, out_dist, out_idx, s, "rf_l2_topk(merge)");
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
// one workgroup per (chunk, k, slot): copies a 16^3 box of a DB scene
__global__ __launch_bounds__(256) void k_gather_patches(const float* __restrict__ vols, long long n_scenes, const int* __restrict__ meta, int K,
                                                        float trunc_fill, float ratio, float mean, float stddev, int layout,
                                                        float* __restrict__ out) {
    const int slot = blockIdx.x & 63, k = (blockIdx.x >> 6) % K, chunk = (blockIdx.x >> 6) / K;
    const int* m = meta + (((size_t)chunk * 64 + slot) * K + k) * 7;
    const int scene = m[0], x0 = m[1], y0 = m[3], z0 = m[5];
    const bool have = scene >= 0 && scene < n_scenes;
    const float* src = vols + (size_t)(have ? scene : 0) * 64 * 64 * 64;
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
    for (int i = threadIdx.x; i < 4096; i += 256) {
        const int dz = i & 15, dy = (i >> 4) & 15, dx = i >> 8;
        float v = trunc_fill;
        if (have) v = src[((size_t)(x0 + dx) * 64 + (y0 + dy)) * 64 + (z0 + dz)];
        v = __fmul_rn(v, ratio);
        dst[dx * dsx + dy * dsy + dz] = __fdiv_rn(__fsub_rn(v, mean), stddev);
    }
}

extern "C" int rf_gather_patches(const float* db_volumes, int64_t n_scenes, const int32_t* meta, int chunks, int K,
                                 float trunc_fill, float trunc_ratio, float mean, float stddev, int layout,
                                 float* out, void* stream) {
    RF_REQUIRE(db_volumes && meta && out && chunks > 0 && K > 0 && n_scenes > 0, RF_E_INVALID, "rf_gather_patches: bad arguments");
    RF_REQUIRE(layout == 0 || layout == 1, RF_E_INVALID, "rf_gather_patches: layout %d", layout);
    hipLaunchKernelGGL(k_gather_patches, dim3((unsigned)chunks * K * 64), dim3(256), 0, (hipStream_t)stream, db_volumes, (long long)n_scenes, meta,
                       K, trunc_fill, trunc_ratio, mean, stddev, layout, out);
    RF_CHECK_LAUNCH("rf_gather_patches");
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
