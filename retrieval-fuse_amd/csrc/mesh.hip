// Iso-surface extraction for the scene-level output of the path: the reference's inference loop ends in
// util/visualization.py:34-37 visualize_sdf_as_mesh = marching_cubes(sdf, level = 0.75) -> .obj (trainer/train_refinement.py:170-173, through
// dataset/scene.py's visualize_*_chunk).  The third-party `marching_cubes` package is not in the reference tree and not pinned: PARITY UNPINNED --
// the contract here is the marching-cubes construction itself (vertices on the grid edges the level crosses, linear interpolation, one closed oriented
// polygon per surface loop of a cube), checked by invariants and against the independent restatement in oracle/mesh.py.
//
// Welded output: a vertex belongs to a GRID EDGE (x, y, z, axis), so it is created once and every cube that touches the edge refers to the same index.
//   pass 1 (rf_mc_classify): per cube the corner configuration -> number of triangles; per grid edge whether the level crosses it
//   (exclusive scans of both on the caller's side: torch.cumsum)
//   pass 2 (rf_mc_emit): vertices of the crossed edges at their scanned index; triangles of every cube at its scanned offset, corners looked up by edge
// HBM-bound index work: 4 B read per voxel and pass, 12 B written per vertex / triangle.  Volumes are [X][Y][Z] fp32 (z fastest), "inside" = value < level.
#include "common.h"

namespace {
// corner c = (cx, cy, cz) bits 0..2; cube edge e = axis * 4 + (u + 2 v) with (u, v) the corner coordinates on the other two axes in ascending axis order
__device__ __forceinline__ void edge_origin(int e, int& ax, int& dx, int& dy, int& dz) {
    ax = e >> 2;
    const int u = e & 1, v = (e >> 1) & 1;
    dx = ax == 0 ? 0 : u;
    dy = ax == 1 ? 0 : (ax == 0 ? u : v);
    dz = ax == 2 ? 0 : v;
}
}   // namespace

__global__ __launch_bounds__(256) void k_mc_classify(const float* __restrict__ sdf, int X, int Y, int Z, float level, const signed char* __restrict__ tri_count,
                                                     int* __restrict__ cube_ntri, int* __restrict__ edge_flag) {
    const size_t total = (size_t)X * Y * Z;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int z = (int)(i % Z), y = (int)((i / Z) % Y), x = (int)(i / ((size_t)Z * Y));
        const bool in0 = sdf[i] < level;
        // the three grid edges that start at this voxel
        edge_flag[i * 3 + 0] = (x + 1 < X) && (in0 != (sdf[i + (size_t)Y * Z] < level));
        edge_flag[i * 3 + 1] = (y + 1 < Y) && (in0 != (sdf[i + Z] < level));
        edge_flag[i * 3 + 2] = (z + 1 < Z) && (in0 != (sdf[i + 1] < level));
        int nt = 0;
        if (x + 1 < X && y + 1 < Y && z + 1 < Z) {
            int cfg = 0;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                cfg |= (sdf[i + (size_t)(c & 1) * Y * Z + (size_t)((c >> 1) & 1) * Z + ((c >> 2) & 1)] < level) << c;
            nt = tri_count[cfg];
        }
        cube_ntri[i] = nt;                                              // (cubes are indexed like their corner 0; the last slab of every axis holds none)
    }
}

__global__ __launch_bounds__(256) void k_mc_emit(const float* __restrict__ sdf, int X, int Y, int Z, float level, const signed char* __restrict__ tri_table,
                                                 const long long* __restrict__ cube_off, const long long* __restrict__ edge_off,
                                                 const int* __restrict__ cube_ntri, const int* __restrict__ edge_flag,
                                                 float* __restrict__ verts, int* __restrict__ tris) {
    const size_t total = (size_t)X * Y * Z;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int z = (int)(i % Z), y = (int)((i / Z) % Y), x = (int)(i / ((size_t)Z * Y));
        const float v0 = sdf[i];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (edge_flag[i * 3 + a]) {
                const float v1 = sdf[i + (a == 0 ? (size_t)Y * Z : (a == 1 ? (size_t)Z : 1))];
                const float t = (level - v0) / (v1 - v0);
                float* p = verts + (size_t)edge_off[i * 3 + a] * 3;
                p[0] = (float)x + (a == 0 ? t : 0.f);
                p[1] = (float)y + (a == 1 ? t : 0.f);
                p[2] = (float)z + (a == 2 ? t : 0.f);
            }
        }
        const int nt = cube_ntri[i];
        if (nt) {
            int cfg = 0;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                cfg |= (sdf[i + (size_t)(c & 1) * Y * Z + (size_t)((c >> 1) & 1) * Z + ((c >> 2) & 1)] < level) << c;
            int* out = tris + (size_t)cube_off[i] * 3;
            for (int k = 0; k < 3 * nt; ++k) {
                int ax, dx, dy, dz;
                edge_origin(tri_table[cfg * 16 + k], ax, dx, dy, dz);
                const size_t g = (((size_t)(x + dx) * Y + (y + dy)) * Z + (z + dz)) * 3 + ax;
                out[k] = (int)edge_off[g];
            }
        }
    }
}

extern "C" int rf_mc_classify(const float* sdf, int x, int y, int z, float level, const signed char* tri_count, int* cube_ntri, int* edge_flag, void* stream) {
    RF_REQUIRE(sdf && tri_count && cube_ntri && edge_flag && x >= 2 && y >= 2 && z >= 2 && (size_t)x * y * z * 3 < (1ull << 31), RF_E_INVALID,
               "rf_mc_classify: needs a volume of at least 2^3 and fewer than 2^31 / 3 voxels (got %d x %d x %d)", x, y, z);
    const size_t want = ((size_t)x * y * z + 255) / 256;
    hipLaunchKernelGGL(k_mc_classify, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0, (hipStream_t)stream, sdf, x, y, z, level, tri_count, cube_ntri, edge_flag);
    RF_CHECK_LAUNCH("rf_mc_classify");
    return RF_OK;
}

extern "C" int rf_mc_emit(const float* sdf, int x, int y, int z, float level, const signed char* tri_table, const long long* cube_off, const long long* edge_off,
                          const int* cube_ntri, const int* edge_flag, float* verts, int* tris, void* stream) {
    RF_REQUIRE(sdf && tri_table && cube_off && edge_off && cube_ntri && edge_flag && verts && tris && x >= 2 && y >= 2 && z >= 2, RF_E_INVALID, "rf_mc_emit: bad arguments");
    const size_t want = ((size_t)x * y * z + 255) / 256;
    hipLaunchKernelGGL(k_mc_emit, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0, (hipStream_t)stream, sdf, x, y, z, level, tri_table, cube_off, edge_off,
                       cube_ntri, edge_flag, verts, tris);
    RF_CHECK_LAUNCH("rf_mc_emit");
    return RF_OK;
}


// ------------------------------------------------------------------------------------------------ scene recomposition
// combine_predictions on the device (reference dataset/patched_scene_dataset.py:160-186: a float64 canvas, refined 64^3 chunks pasted at their origins; the
// predictions went through `.cpu().half()` first, trainer/train_refinement.py:160-167): m chunks of a refined batch df [b][64^3] fp32 -> rounded to float16
// (round to nearest even, as torch's .half()) -> widened to float64 (exact) -> canvas rows.  sel[i] = the chunk's index in the batch, dst[i] = the element
// offset of its origin in `flat`, sx / sy = the canvas' strides of x and y in elements (z is contiguous).  One workgroup per (chunk, x): a 64 x 64 (y, z)
// plane, rows of 64 doubles = 512 contiguous bytes.  Replaces df.half() -> .double() -> gather -> index_put (four passes over 67 MB per 32 chunks).
__global__ __launch_bounds__(256) void k_paste_chunks(const float* __restrict__ df, const int* __restrict__ sel, const long long* __restrict__ dst, long long sx,
                                                      long long sy, int round_half, double* __restrict__ flat) {
    const int c = blockIdx.x >> 6, x = blockIdx.x & 63;
    const float4* __restrict__ src = reinterpret_cast<const float4*>(df + ((size_t)sel[c] * 64 + x) * 4096);
    double* __restrict__ out = flat + dst[c] + (long long)x * sx;
    for (int q = threadIdx.x; q < 1024; q += 256) {                  // float4 q: (y, z) = (q >> 4, 4 (q & 15))
        float4 v = src[q];
        if (round_half) { v.x = (float)(_Float16)v.x; v.y = (float)(_Float16)v.y; v.z = (float)(_Float16)v.z; v.w = (float)(_Float16)v.w; }
        double* o = out + (long long)(q >> 4) * sy + 4 * (q & 15);
        *reinterpret_cast<double2*>(o) = make_double2((double)v.x, (double)v.y);
        *reinterpret_cast<double2*>(o + 2) = make_double2((double)v.z, (double)v.w);
    }
}

extern "C" int rf_paste_chunks(const float* df, int b, const int* sel, const long long* dst, int m, long long sx, long long sy, int round_half, double* flat,
                               void* stream) {
    RF_REQUIRE(df && sel && dst && flat && b > 0 && m >= 0, RF_E_INVALID, "rf_paste_chunks: bad arguments");
    RF_REQUIRE(sy >= 64 && sx >= 64 * sy && sy % 2 == 0 && sx % 2 == 0, RF_E_INVALID, "rf_paste_chunks: canvas strides sx=%lld sy=%lld (need even, sy >= 64, sx >= 64 sy)", sx, sy);
    if (m == 0) return RF_OK;
    hipLaunchKernelGGL(k_paste_chunks, dim3((unsigned)m * 64u), dim3(256), 0, (hipStream_t)stream, df, sel, dst, sx, sy, round_half, flat);
    RF_CHECK_LAUNCH("rf_paste_chunks");
    return RF_OK;
}
