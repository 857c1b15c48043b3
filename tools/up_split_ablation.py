"""Dev tool: where does the dominant launch (k_conv3_up_split<4>, 32+64->56 @8^3 x 8192) spend a box's life?  Builds one-patch variants of
csrc/conv3d_up_split.hip (CPU container: `python tools/up_split_ablation.py build`) and times them on the GPU box (`python tools/up_split_ablation.py`);
the `stamps` variant records s_memtime at the phase borders of every workgroup (wave 0)."""
import ctypes, subprocess, sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
CSRC = REPO / 'retrieval-fuse_amd' / 'csrc'
OUT = REPO / 'tools' / '_haz'

MFMA = """    for (int n = 0; n < NB; ++n) hi[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[n], hi[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NB; ++n) lo[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[n], lo[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NB; ++n) lo[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[n], lo[n], 0, 0, 0);
}

template <int NB>
__global__ __launch_bounds__(512, 2) void k_conv3_up_split("""
NO_MFMA = """    for (int n = 0; n < NB; ++n) { hi[n][0] += (float)ah[0] * (float)bh[n][0]; lo[n][0] += (float)al[0] * (float)bl[n][0]; }
}

template <int NB>
__global__ __launch_bounds__(512, 2) void k_conv3_up_split("""
ZERO = "    for (int i = tid; i < US_LDS_BYTES / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0u, 0u, 0u, 0u);\n    __syncthreads();\n\n    // ---- stage: all low-res"
NO_ZERO = "    __syncthreads();\n\n    // ---- stage: all low-res"
XLOAD = "            for (int j = 0; j < 8; ++j) x[j] = s0[(size_t)(cx * 8 + j) * 512];\n        };\n        const unsigned char* buf = lds + (ca & 1) * US_A_BUF + abase;"
NO_XLOAD = "            for (int j = 0; j < 8; ++j) x[j] = (float)(tid + j + cx);\n        };\n        const unsigned char* buf = lds + (ca & 1) * US_A_BUF + abase;"
BLOAD = """            bh[nb] = wn[(nb * 2) * 64];
            bl[nb] = wn[(nb * 2 + 1) * 64];
        }
        wn += STEP_U4;
    };
    load_b(b0h, b0l);

    h8 ah[2], al[2];
    __syncthreads();

    // one k-step: the next step's B fragments are requested first (a full step ahead"""
NO_BLOAD = """            bh[nb] = h8{(_Float16)(float)(lane + nb), 0, 0, 0, 0, 0, 0, 0};
            bl[nb] = h8{(_Float16)(float)nb, 0, 0, 0, 0, 0, 0, 0};
            asm volatile("" : "+v"(bh[nb]), "+v"(bl[nb]));
        }
        wn += STEP_U4;
    };
    load_b(b0h, b0l);

    h8 ah[2], al[2];
    __syncthreads();

    // one k-step: the next step's B fragments are requested first (a full step ahead"""
STORE = "        *reinterpret_cast<float4*>(o + (size_t)co * 512 + l4 * 4) = *reinterpret_cast<const float4*>(e + co * US_E_STRIDE + l4 * 4);\n    }\n    if (a.stats) {\n        // per cout: eight threads sum 64 values"
NO_STORE = "        if (e[co * US_E_STRIDE + l4 * 4] == 123.456f) *reinterpret_cast<float4*>(o + (size_t)co * 512 + l4 * 4) = *reinterpret_cast<const float4*>(e + co * US_E_STRIDE + l4 * 4);\n    }\n    if (a.stats) {\n        // per cout: eight threads sum 64 values"
EPI_HEAD = "    // ---- epilogue: out = relu(hi + lo / 2^11) -> LDS tile [cout][z][y][x] -> float4 rows\n    float* e = reinterpret_cast<float*>(lds);\n    {"
NO_EPI_HEAD = """    // ---- epilogue: out = relu(hi + lo / 2^11) -> LDS tile [cout][z][y][x] -> float4 rows
    float* e = reinterpret_cast<float*>(lds);
    {
        float sink = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) sink += hi[m][nb][r] + lo[m][nb][r];
        if (sink == 123.456f) a.out[tid] = sink;
        return;
    }
    {"""
CONV = """        h8 h, l;
        us_split8(y, h, l);
        unsigned char* p = lds + (ca & 1) * US_A_BUF + vslot * 16;"""
NO_CONV = """        h8 h, l;
        for (int j = 0; j < 8; ++j) { h[j] = (_Float16)x[j]; l[j] = (_Float16)y[j]; }
        unsigned char* p = lds + (ca & 1) * US_A_BUF + vslot * 16;"""

# ---- stamps: wave 0 lane 0 of each workgroup writes s_memtime at phase borders
ST_DECL = "template <int NB>\n__global__ __launch_bounds__(512, 2) void k_conv3_up_split(UpSplitArgs a) {\n    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];"
ST_DECL_NEW = """__device__ unsigned long long g_stamps[8192 * 12];
extern "C" int rft_read_stamps(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8192 * 12); }
#define STAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (threadIdx.x == STW * 64 && blockIdx.x < 8192) g_stamps[blockIdx.x * 12 + (i)] = t_; } while (0)
template <int NB>
__global__ __launch_bounds__(512, 2) void k_conv3_up_split(UpSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    STAMP(0);"""
ST = [
    (ST_DECL, ST_DECL_NEW),
    ("    __syncthreads();\n\n    // ---- stage: all low-res", "    __syncthreads();\n    STAMP(1);\n\n    // ---- stage: all low-res"),
    ("    h8 ah[2], al[2];\n    __syncthreads();\n", "    h8 ah[2], al[2];\n    STAMP(2);\n    __syncthreads();\n    STAMP(3);\n"),
    ("    // ---- phase B: upsampled channels in low resolution\n    {", "    STAMP(4);\n    // ---- phase B: upsampled channels in low resolution\n    {"),
    ("    __syncthreads();\n\n    // ---- epilogue: out = relu", "    STAMP(5);\n    __syncthreads();\n    STAMP(6);\n\n    // ---- epilogue: out = relu"),
    ("    __syncthreads();\n    const int cout = a.cout;\n    float* __restrict__ o = a.out + (size_t)n * cout * 512;", "    STAMP(7);\n    __syncthreads();\n    STAMP(8);\n    const int cout = a.cout;\n    float* __restrict__ o = a.out + (size_t)n * cout * 512;"),
    ("    if (a.stats) {\n        // per cout: eight threads sum 64 values", "    STAMP(9);\n    if (a.stats) {\n        // per cout: eight threads sum 64 values"),
    ("        if (part == 0 && co < cout) a.stats[(size_t)n * cout + co] = make_double2(sm, sq);\n    }\n    };   // run",
     "        if (part == 0 && co < cout) a.stats[(size_t)n * cout + co] = make_double2(sm, sq);\n    }\n    STAMP(10);\n    };   // run"),
]
SW = 'constexpr bool US_ZSKIP = true, US_PERSISTENT = true;'
VARIANTS = {'base': [], 'no_zskip': [(SW, SW.replace('US_ZSKIP = true', 'US_ZSKIP = false'))], 'one_sample': [(SW, SW.replace('US_PERSISTENT = true', 'US_PERSISTENT = false'))], 'no_shadow': [(SW, SW.replace('US_SHADOW = true', 'US_SHADOW = false'))],
            'no_epilogue': [(EPI_HEAD, NO_EPI_HEAD)], 'no_conv': [(CONV, NO_CONV)], 'no_xload_no_bload': [(XLOAD, NO_XLOAD), (BLOAD, NO_BLOAD)],
            'mfma_only': [(XLOAD, NO_XLOAD), (BLOAD, NO_BLOAD), (EPI_HEAD, NO_EPI_HEAD), (ZERO, NO_ZERO), (CONV, NO_CONV)],
            'stamps0': [(a, b.replace('STW', '0')) for a, b in ST], 'stamps7': [(a, b.replace('STW', '7')) for a, b in ST]}

# ---- stamps inside the phase-A chunks 2 and 3, kept in SGPRs until the end of the kernel (no stores, no branches inside the loop)
def TS(k):
    return "        __builtin_amdgcn_sched_barrier(0); ts[decltype(TAG)::value][%d] = (unsigned)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);\n" % k
KS = [
    ("template <int NB>\n__global__ __launch_bounds__(512, 2) void k_conv3_up_split(UpSplitArgs a) {",
     "__device__ unsigned g_ts[8192 * 8 * 20];\nextern \"C\" int rft_read_ts(unsigned* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ts), sizeof(unsigned) * 8192 * 8 * 20); }\ntemplate <int NB>\n__global__ __launch_bounds__(512, 2) void k_conv3_up_split(UpSplitArgs a) {"),
    ("    auto chunk_a = [&](int ca, h8 (&ch)[NB],", "    unsigned ts[2][10];\n    auto chunk_a = [&](auto TAG, int ca, h8 (&ch)[NB],"),
    ("        chunk_a(ca, b0h, b0l, b1h, b1l);", "        chunk_a(std::integral_constant<int, 0>{}, ca, b0h, b0l, b1h, b1l);"),
    ("        if (ca + 1 < nA) chunk_a(ca + 1, b1h, b1l, b0h, b0l);", "        if (ca + 1 < nA) chunk_a(std::integral_constant<int, 1>{}, ca + 1, b1h, b1l, b0h, b0l);"),
    ("        kstep(skip_lo{}, std::true_type{}, xload_a, buf + atap[0]", TS(0) + "        kstep(skip_lo{}, std::true_type{}, xload_a, buf + atap[0]"),
    ("        kstep(skip_lo{}, std::true_type{}, no_x, buf + atap[1]", TS(1) + "        kstep(skip_lo{}, std::true_type{}, no_x, buf + atap[1]"),
    ("        kstep(no_skip{}, std::true_type{}, no_x, buf + atap[2]", TS(2) + "        kstep(no_skip{}, std::true_type{}, no_x, buf + atap[2]"),
    ("        convert_store(0);\n", TS(3) + "        convert_store(0);\n"),
    ("        kstep(no_skip{}, std::true_type{}, no_x, buf + atap[4]", TS(4) + "        kstep(no_skip{}, std::true_type{}, no_x, buf + atap[4]"),
    ("        kstep(skip_hi{}, std::true_type{}, no_x, buf + atap[5]", TS(5) + "        kstep(skip_hi{}, std::true_type{}, no_x, buf + atap[5]"),
    ("        wn = more ? wn : wB; ", TS(6) + "        wn = more ? wn : wB; "),
    ("        convert_store(1);\n        __syncthreads();\n    };", TS(7) + "        convert_store(1);\n" + TS(8) + "        __syncthreads();\n" + TS(9) + "    };"),
    ("        if (part == 0 && co < cout) a.stats[(size_t)n * cout + co] = make_double2(sm, sq);\n    }\n    };   // run",
     "        if (part == 0 && co < cout) a.stats[(size_t)n * cout + co] = make_double2(sm, sq);\n    }\n    if (lane == 0) for (int q = 0; q < 20; ++q) g_ts[(blockIdx.x * 8 + wave) * 20 + q] = ts[q / 10][q % 10];\n    };   // run"),
]
VARIANTS.update({'ks': list(KS)})
KS_LABELS = ['k-step 0 (requests half A)', 'k-step 1', 'k-step 2', 'convert half A + k-step 3 (requests half B)', 'k-step 4', 'k-step 5', 'k-step 6', 'convert half B', 'barrier']

# ---- latency of the next-chunk voxel loads: issue, wait at once, stamp (the pipeline is destroyed; only the two stamps matter)
XL = [
    (ST_DECL, ST_DECL_NEW),
    ("#pragma unroll\n            for (int j = 0; j < 8; ++j) x[j] = s0[(size_t)(cx * 8 + j) * 512];\n        };",
     "            asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n            STAMP(2 * ca);\n#pragma unroll\n            for (int j = 0; j < 8; ++j) x[j] = s0[(size_t)(cx * 8 + j) * 512];\n            asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n            STAMP(2 * ca + 1);\n        };"),
]
WARM = (SW, SW.replace('US_WARM = false', 'US_WARM = true'))
VARIANTS.update({'xlat': [(a, b.replace('STW', '0')) for a, b in XL] , 'xlat_warm': [(a, b.replace('STW', '0')) for a, b in XL] + [(SW, SW.replace('US_WARM = false', 'US_WARM = true'))]})

# ---- persistent form: stamps per sample (SGPRs, stored by lane 0 of each wave at the end of the sample)
def PS(k):
    return "        __builtin_amdgcn_sched_barrier(0); pts[%d] = (unsigned)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);\n" % k
PST = [
    ("template <int NB>\n__global__ __launch_bounds__(512, 2) void k_conv3_up_split_p(UpSplitArgs a) {",
     "__device__ unsigned g_pts[8192 * 8 * 8];\nextern \"C\" int rft_read_pts(unsigned* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pts), sizeof(unsigned) * 8192 * 8 * 8); }\ntemplate <int NB>\n__global__ __launch_bounds__(512, 2) void k_conv3_up_split_p(UpSplitArgs a) {"),
    ("        f32x4 hi[4][NB], lo[4][NB];\n        h8 ah[2], al[2];\n", "        unsigned pts[8];\n" + PS(0) + "        f32x4 hi[4][NB], lo[4][NB];\n        h8 ah[2], al[2];\n"),
    ("        // ---- phase B: upsampled channels in low resolution\n        {\n            int bq", PS(1) + "        // ---- phase B: upsampled channels in low resolution\n        {\n            int bq"),
    ("        // ---- the next sample's low-res voxels are requested now", PS(2) + "        // ---- the next sample's low-res voxels are requested now"),
    ("        // ---- epilogue: out = relu(hi + lo / 2^11) -> LDS tile [cout][z][y][x] behind buffer 0", PS(3) + "        // ---- epilogue: out = relu(hi + lo / 2^11) -> LDS tile [cout][z][y][x] behind buffer 0"),
    ("        float* __restrict__ o = a.out + (size_t)box * cout * 512;\n        for (int q = t;", PS(4) + "        float* __restrict__ o = a.out + (size_t)box * cout * 512;\n        for (int q = t;"),
    ("        // ---- the tile is dead: buffer 1 back to zeros", PS(5) + "        // ---- the tile is dead: buffer 1 back to zeros"),
    ("        __syncthreads();\n        tb = tn;\n", "        __syncthreads();\n" + PS(6) + "        if ((t & 63) == 0) for (int q = 0; q < 7; ++q) g_pts[(box * 8 + wave) * 8 + q] = pts[q];\n        tb = tn;\n"),
]
VARIANTS.update({'pstamps': list(PST)})
PST_LABELS = ['phase A (4 chunks x 7 k-steps)', 'phase B (8 chunks x 2 k-steps)', 'request next low-res + barrier', 'accumulators -> LDS tile + barrier', 'tile -> stores, statistics, barrier', 're-zero, next low-res, barrier']


def build(only=None):
    OUT.mkdir(exist_ok=True)
    for name, patches in VARIANTS.items():
        if only and name not in only:
            continue
        src = (CSRC / 'conv3d_up_split.hip').read_text()
        for old, new in patches:
            assert src.count(old) == 1, (name, src.count(old), old[:80])
            src = src.replace(old, new)
        p = OUT / ('upsplit_%s.hip' % name)
        p.write_text(src)
        obj = OUT / ('upsplit_%s.o' % name)
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', str(CSRC), '-c', str(p), '-o', str(obj)], check=True)
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(OUT / ('libupsplit_%s.so' % name)), str(obj), str(CSRC / 'build' / 'capi.o')], check=True)
        print(name, flush=True)


def run():
    import numpy as np
    import torch
    sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
    from rfuse import ops
    dev = torch.device('cuda:0')
    VP = ctypes.c_void_p
    n, c0, c1, edge, cout = 8192, 32, 64, 8, 56
    s0 = torch.rand(n, c0, edge, edge, edge, device=dev)
    s1 = torch.rand(n, c1, edge // 2, edge // 2, edge // 2, device=dev)
    w = torch.randn(cout, c0 + c1, 3, 3, 3, device=dev) * 0.05
    aff = torch.zeros(n, c0 + c1, 4, device=dev); aff[..., 1] = 1.0
    ws = ops.pack_conv3_up_split_weight(w, c0)
    out = torch.empty(n, cout, edge, edge, edge, device=dev)
    stats = torch.empty(n, cout, 2, dtype=torch.float64, device=dev)
    names = [a for a in sys.argv[1:]] or ['old'] + list(VARIANTS)
    names = [nm for nm in names if (OUT / ('libupsplit_%s.so' % nm)).exists()]
    st = torch.cuda.current_stream().cuda_stream
    libs = {}
    for name in names:
        lib = ctypes.CDLL(str(OUT / ('libupsplit_%s.so' % name)))
        lib.rf_conv3d_up_split_k3_gn_relu.argtypes = [VP, ctypes.c_int, VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, VP, VP, ctypes.c_int, VP, VP, VP]
        libs[name] = lib
    def call(name, with_stats=True):
        return libs[name].rf_conv3d_up_split_k3_gn_relu(s0.data_ptr(), c0, s1.data_ptr(), c1, n, edge, aff.data_ptr(), ws.data_ptr(), cout, out.data_ptr(), stats.data_ptr() if with_stats else None, st)
    for _ in range(300):                                           # clocks and caches settle
        call(names[0])
    torch.cuda.synchronize()
    times = {(nm, ws_): [] for nm in names for ws_ in (True, False)}
    for rnd in range(5):                                           # variants interleaved: box-to-box and minute-to-minute drift hits all alike
        for name in names:
            for with_stats in (True, False):
                for _ in range(2):
                    assert call(name, with_stats) == 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    call(name, with_stats)
                e1.record(); torch.cuda.synchronize()
                times[(name, with_stats)].append(e0.elapsed_time(e1) * 100)
    for name in names:
        a1, a0 = sorted(times[(name, True)]), sorted(times[(name, False)])
        print('%-22s with stats %8.1f us (min %8.1f)   without %8.1f us (min %8.1f)' % (name, a1[len(a1) // 2], a1[0], a0[len(a0) // 2], a0[0]), flush=True)
    for name in names:
        lib = libs[name]
        f = lib.rf_conv3d_up_split_k3_gn_relu
        if name.startswith('xlat'):
            lib.rft_read_stamps.argtypes = [VP]
            buf = np.zeros((8192, 12), dtype=np.uint64)
            f(s0.data_ptr(), c0, s1.data_ptr(), c1, n, edge, aff.data_ptr(), ws.data_ptr(), cout, out.data_ptr(), stats.data_ptr(), st)
            torch.cuda.synchronize()
            assert lib.rft_read_stamps(buf.ctypes.data) == 0
            t = buf.astype(np.int64)
            for ca in range(4):
                d = t[:, 2 * ca + 1] - t[:, 2 * ca]
                print('  %s: chunk %d requests its successor: issue -> all 8 loads landed: mean %.0f median %.0f p10 %.0f p90 %.0f ticks' % (name, ca, d.mean(), np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
        if name.startswith('pstamps'):
            lib.rft_read_pts.argtypes = [VP]
            buf = np.zeros((8192, 8, 8), dtype=np.uint32)
            f(s0.data_ptr(), c0, s1.data_ptr(), c1, n, edge, aff.data_ptr(), ws.data_ptr(), cout, out.data_ptr(), stats.data_ptr(), st)
            torch.cuda.synchronize()
            assert lib.rft_read_pts(buf.ctypes.data) == 0
            d = np.diff(buf[:, :, :7].astype(np.int64), axis=-1) % (1 << 32)
            for wv in (0, 4):
                print('  %s, wave %d: ticks per sample (samples of the steady state)' % (name, wv))
                sel = d[512:7680, wv]
                for i, lab in enumerate(PST_LABELS):
                    print('    %-42s %8.0f (p90 %6.0f)  %5.1f %%' % (lab, sel[:, i].mean(), np.percentile(sel[:, i], 90), 100 * sel[:, i].mean() / sel.sum(axis=-1).mean()))
                print('    %-42s %8.0f' % ('sample', sel.sum(axis=-1).mean()))
        if name.startswith('ks'):
            lib.rft_read_ts.argtypes = [VP]
            buf = np.zeros((8192, 8, 2, 10), dtype=np.uint32)
            f(s0.data_ptr(), c0, s1.data_ptr(), c1, n, edge, aff.data_ptr(), ws.data_ptr(), cout, out.data_ptr(), stats.data_ptr(), st)
            torch.cuda.synchronize()
            assert lib.rft_read_ts(buf.ctypes.data) == 0
            d = np.diff(buf.astype(np.int64), axis=-1) % (1 << 32)                                # [box, wave, chunk 2|3, 9 intervals]
            for wv in (0, 4):
                print('  %s, wave %d: ticks per interval, chunk 2 (requests chunk 3) | chunk 3 (re-stages itself)' % (name, wv))
                for i, lab in enumerate(KS_LABELS):
                    print('    %-42s %8.0f (p90 %6.0f) | %8.0f' % (lab, d[256:, wv, 0, i].mean(), np.percentile(d[256:, wv, 0, i], 90), d[256:, wv, 1, i].mean()))
                print('    %-42s %8.0f          | %8.0f' % ('chunk', d[256:, wv, 0].sum(axis=-1).mean(), d[256:, wv, 1].sum(axis=-1).mean()))
        if name.startswith('stamps'):
            lib.rft_read_stamps.argtypes = [VP]
            buf = np.zeros((8192, 12), dtype=np.uint64)
            torch.cuda.synchronize()
            f(s0.data_ptr(), c0, s1.data_ptr(), c1, n, edge, aff.data_ptr(), ws.data_ptr(), cout, out.data_ptr(), stats.data_ptr(), st)
            torch.cuda.synchronize()
            assert lib.rft_read_stamps(buf.ctypes.data) == 0
            t = buf[:, :11].astype(np.int64)
            d = np.diff(t, axis=1)
            labels = ['LDS zero-fill + barrier', 'stage low-res + chunk 0, first weights', 'barrier', 'phase A (4 chunks x 7 k-steps)', 'phase B (8 chunks x 2 k-steps)', 'barrier',
                      'accumulators -> LDS tile', 'barrier', 'tile -> float4 stores', 'statistics (float64)']
            if name.startswith('ks'):
                labels = KS_LABELS
            life = (t[:, 10] - t[:, 0])
            print('  ticks per box (mean / median / p90), total life mean %.0f; span of the launch %.0f ticks' % (life.mean(), float(t[:, 10].max() - t[:, 0].min())))
            for i, lab in enumerate(labels):
                print('  %-42s %8.0f %8.0f %8.0f   %5.1f %%' % (lab, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90), 100 * d[:, i].mean() / life.mean()))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        build(sys.argv[2:])
    else:
        run()
