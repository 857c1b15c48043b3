"""Dev tool: where does the multi-chunk persistent z-column kernel (k_conv3_split_zcm, csrc/conv3d_split_zc.hip) spend its time?  One-patch variants,
built in the CPU container (`python tools/zcm_ablation.py build`), timed on the GPU box (`python tools/zcm_ablation.py [variants]`) on the two layers of
the C2 step it takes: 56 -> 16 @8^3 x 8192 (pre-split input, full output) and 16 -> 16 @64^3 x 32 (fp32 input, pointwise head).  Variant results are
wrong on purpose; timing and wave cycles only."""
import ctypes, subprocess, sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
CSRC = REPO / 'retrieval-fuse_amd' / 'csrc'
OUT = REPO / 'tools' / '_haz'
ENTRY = '''
extern "C" int zcm_run(const void* src, int pre, const float* aff, const void* wp, float* out, float* pw_out, const float* pw_w, const float* pw_b, int cin, int n, int edge, void* stream,
                       int cout, float* pool_out, double* stats, double* pool_stats) {
    ConvArgs a;
    a.src0 = reinterpret_cast<const float*>(src); a.src1 = nullptr; a.affine = reinterpret_cast<const float4*>(aff); a.wp = reinterpret_cast<const float*>(wp); a.out = out;
    a.c0 = cin; a.c1 = 0; a.n = n; a.edge = edge; a.cout = cout; a.cin4 = cin; a.cout16 = (cout + 15) / 16 * 16; a.stats = reinterpret_cast<double2*>(stats); a.stats_tiles = 1;
    a.pool_out = pool_out; a.pool_stats = reinterpret_cast<double2*>(pool_stats); a.pool_mode = pool_out ? 1 : 0; a.floor = 0.f;
    const SplitPreOut po{nullptr, nullptr, nullptr, 0, 0.f, pw_out, pw_w, pw_b, 1.0f, 0.5f};
    return rf_split_zcm_launch(a, po, pre != 0, (hipStream_t)stream, "zcm_run");
}
'''
MF = "auto mf = [](const h8& x, const h8& y, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c, 0, 0, 0); };"
NO_MF = "auto mf = [](const h8& x, const h8& y, const f32x4& c) { f32x4 r = c; r[0] += (float)x[0] * (float)y[0]; return r; };"
MID = "            lds_barrier();                                            // MID: R1 in place, R0 free"
LOADS_PRE = """                st.ph[r] = *reinterpret_cast<const h8*>(p + (size_t)off * 16);
                st.pl[r] = *reinterpret_cast<const h8*>(p + (vol + off) * 16);"""
NO_LOADS_PRE = """                st.ph[r] = h8{(_Float16)(float)off, 0, 0, 0, 0, 0, 0, 0};
                st.pl[r] = h8{(_Float16)(float)r, 0, 0, 0, 0, 0, 0, 0};"""
LOADS_F32 = "                for (int j = 0; j < 8; ++j) st.x[r][j] = p[(size_t)j * vol];"
NO_LOADS_F32 = "                for (int j = 0; j < 8; ++j) st.x[r][j] = (float)(off + j);"
CONV = "                for (int j = 0; j < 8; ++j) y[j] = st.in[r] ? fmaf(st.x[r][j] - af[j].x, af[j].y, af[j].z) : 0.f;\n                cs_split8(y, h, l);"
NO_CONV = "                for (int j = 0; j < 8; ++j) { h[j] = (_Float16)st.x[r][j]; l[j] = (_Float16)af[j].y; }"
W1 = "            const h8 w1 = wsrc[(size_t)ca * wstride + wsrc1];        // R1 of this chunk: lands under pass 0"
NO_W1 = "            const h8 w1 = h8{(_Float16)(float)ca, 0, 0, 0, 0, 0, 0, 0};"
W0 = "            const h8 w0 = wsrc[(size_t)nca * wstride + wsrc0];       // R0 of the next chunk: lands under pass 1"
NO_W0 = "            const h8 w0 = h8{(_Float16)(float)nca, 0, 0, 0, 0, 0, 0, 0};"
SST = "            stage_store(st, nbox, nca, other);\n            if (tid < 384)"
NO_SST = "            if (box < 0) stage_store(st, nbox, nca, other);\n            if (tid < 384)"
EPI_ALL = "            if (!last) continue;"
NO_EPI_ALL = "            if (!last || box >= 0) continue;"
EPI_OUT = "                if (a.pool_mode != 2) {\n                    if (co_g < a.cout) {"
NO_EPI_OUT = "                if (a.pool_mode != 2) {\n                    if (co_g < a.cout && hi[0][0] == 123.456f) {"
DECL = "    int item = 0;                                                    // parity of the image buffer"
DECL_STAMP = "    const long long dbg_t0 = __builtin_readcyclecounter();\n    int item = 0;"
END = "                        o[(size_t)sg * 2 * 512 + 512] = l;\n                    }\n                }\n            }\n        }\n    }\n}"
END_STAMP = "                        o[(size_t)sg * 2 * 512 + 512] = l;\n                    }\n                }\n            }\n        }\n    }\n    if (lane == 0 && dbg_out) { dbg_out[blockIdx.x * 8 + wave] = (float)(__builtin_readcyclecounter() - dbg_t0); if (wave == 0) { reinterpret_cast<long long*>(dbg_out + 4096)[(blockIdx.y * 512 + blockIdx.x) * 2] = dbg_t0; reinterpret_cast<long long*>(dbg_out + 4096)[(blockIdx.y * 512 + blockIdx.x) * 2 + 1] = __builtin_readcyclecounter(); } }\n}"
VARIANTS = {'base': [], 'no_mfma': [(MF, NO_MF)], 'no_mid_barrier': [(MID, '')], 'no_loads': [(LOADS_PRE, NO_LOADS_PRE), (LOADS_F32, NO_LOADS_F32)],
            'no_conversion': [(CONV, NO_CONV)], 'no_weight_loads': [(W1, NO_W1), (W0, NO_W0)], 'no_staging': [(LOADS_PRE, NO_LOADS_PRE), (LOADS_F32, NO_LOADS_F32), (SST, NO_SST)],
            'no_epilogue': [(EPI_ALL, NO_EPI_ALL), (EPI_OUT, NO_EPI_OUT)],
            'compute_only': [(LOADS_PRE, NO_LOADS_PRE), (LOADS_F32, NO_LOADS_F32), (SST, NO_SST), (EPI_ALL, NO_EPI_ALL), (EPI_OUT, NO_EPI_OUT), (W1, NO_W1), (W0, NO_W0)],
            'compute_only_no_mid': [(LOADS_PRE, NO_LOADS_PRE), (LOADS_F32, NO_LOADS_F32), (SST, NO_SST), (EPI_ALL, NO_EPI_ALL), (EPI_OUT, NO_EPI_OUT), (W1, NO_W1), (W0, NO_W0), (MID, '')]}


def build(names=None):
    OUT.mkdir(exist_ok=True)
    for name, patches in VARIANTS.items():
        if names and name not in names:
            continue
        src = (CSRC / 'conv3d_split_zc.hip').read_text()
        src = src.replace('__global__ __launch_bounds__(512, 4) void k_conv3_split_zcm(ConvArgs a, SplitPreOut po, int boxes_per_wg, int total_boxes) {',
                          '__global__ __launch_bounds__(512, 4) void k_conv3_split_zcm(ConvArgs a, SplitPreOut po, int boxes_per_wg, int total_boxes) {\n    float* dbg_out = reinterpret_cast<float*>(const_cast<float*>(a.src1));')
        cut = src.index('// ------------------------------------------------------------------------------------------------- multi-chunk layers')
        head, src = src[:cut], src[cut:]
        for old, new in list(patches) + [(DECL, DECL_STAMP), (END, END_STAMP)]:
            assert src.count(old) == 1, (name, old[:70], src.count(old))
            src = src.replace(old, new)
        src = head + src
        p = OUT / ('zcm_%s.hip' % name)
        p.write_text(src + ENTRY.replace('a.src1 = nullptr;', 'a.src1 = pw_b ? nullptr : nullptr; a.src1 = g_dbg;').replace('extern "C" int zcm_run', 'static float* g_dbg = nullptr;\nextern "C" void zcm_dbg(float* p) { g_dbg = p; }\nextern "C" int zcm_run'))
        obj = OUT / ('zcm_%s.o' % name)
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', str(CSRC), '-c', str(p), '-o', str(obj)], check=True)
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(OUT / ('libzcm_%s.so' % name)), str(obj), str(CSRC / 'build' / 'capi.o')], check=True)
        print(name)


def run():
    import torch
    sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
    from rfuse import ops
    dev = torch.device('cuda:0')
    VP = ctypes.c_void_p
    names = [a for a in sys.argv[1:]] or list(VARIANTS)
    for label, pre, cin, n, edge, cout in (('56->16 @8^3 x 8192, pre-split in, full out', 1, 56, 8192, 8, 16), ('16->16 @64^3 x 32, fp32 in, pointwise head', 0, 16, 32, 64, 16),
                                            ('16->32 @8^3 x 8192, pre-split in, full + pooled out, statistics', 1, 16, 8192, 8, 32)):
        if pre:
            src = torch.randint(0, 255, (n * (cin // 8) * 2 * edge ** 3 * 16,), dtype=torch.uint8, device=dev)
            src.view(torch.float16).clamp_(-4, 4); src.view(torch.float16).nan_to_num_(0.0)
        else:
            src = torch.randn(n, cin, edge, edge, edge, device=dev).relu_()
        aff = torch.zeros(n, cin, 4, device=dev); aff[..., 1] = 1.0
        w = ops.pack_conv3_split_weight(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05)
        out = torch.empty(n, cout, edge, edge, edge, device=dev)
        pooled = torch.empty(n, cout, edge // 2, edge // 2, edge // 2, device=dev) if cout == 32 else None
        st1 = torch.empty(n, cout, 1, 2, dtype=torch.float64, device=dev) if cout == 32 else None
        st2 = torch.empty(n, cout, 1, 2, dtype=torch.float64, device=dev) if cout == 32 else None
        pw_out = torch.empty(n, 1, edge, edge, edge, device=dev); pw_w = torch.randn(16, device=dev); pw_b = torch.zeros(1, device=dev)
        dbg = torch.zeros(4096 + 4 * 2048, device=dev)
        print(label)
        for name in names:
            so = OUT / ('libzcm_%s.so' % name)
            if not so.exists():
                continue
            lib = ctypes.CDLL(str(so))
            lib.zcm_dbg.argtypes = [VP]; lib.zcm_dbg(dbg.data_ptr())
            f = lib.zcm_run
            f.argtypes = [VP, ctypes.c_int, VP, VP, VP, VP, VP, VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, VP, ctypes.c_int, VP, VP, VP]
            st = torch.cuda.current_stream().cuda_stream
            call = lambda: f(src.data_ptr(), pre, aff.data_ptr(), w.data_ptr(), out.data_ptr(), None if (pre or cout == 32) else pw_out.data_ptr(), pw_w.data_ptr(), pw_b.data_ptr(), cin, n, edge, st,
                             cout, pooled.data_ptr() if pooled is not None else None, st1.data_ptr() if st1 is not None else None, st2.data_ptr() if st2 is not None else None)
            for _ in range(5): assert call() == 0, name
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): call()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 50
            cyc = dbg[:4096][dbg[:4096] > 0].mean().item()
            ts = dbg[4096:].view(torch.int64).view(-1, 2).cpu()
            ts = ts[ts[:, 0] > 0]
            span = (ts[:, 1].max() - ts[:, 0].min()).item()
            late = (ts[:, 0] > ts[:, 0].min() + 0.25 * span).sum().item()
            print('   %-22s %8.1f us   %8.0f k cycles per wave, kernel span %8.0f k cycles (%.2f GHz); %d of %d workgroups started after the first quarter' % (name, us, cyc / 1e3, span / 1e3, span / us / 1e3, late, len(ts)), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        build(sys.argv[2:])
    else:
        run()
