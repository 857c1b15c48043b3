"""Dev tool: where does the persistent z-column kernel (csrc/conv3d_split_zc.hip; 8->16 @16^3 x 8192, pooled-only) spend its time?  Builds one-patch
variants (CPU container: `python tools/zc_ablation.py build`) and times them on the GPU box (`python tools/zc_ablation.py`).  Variant results are
wrong on purpose; timing only."""
import ctypes, subprocess, sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
CSRC = REPO / 'retrieval-fuse_amd' / 'csrc'
OUT = REPO / 'tools' / '_haz'
ENTRY = '''
extern "C" int zc_run(const void* pre, const void* wp, float* pool_out, double* pstats, int n, void* stream, float* dbg) {
    ConvArgs a;
    a.src0 = reinterpret_cast<const float*>(pre); a.src1 = nullptr; a.affine = nullptr; a.wp = reinterpret_cast<const float*>(wp); a.out = dbg;
    a.c0 = 8; a.c1 = 0; a.n = n; a.edge = 16; a.cout = 16; a.cin4 = 8; a.cout16 = 16; a.stats = nullptr; a.stats_tiles = 1;
    a.pool_out = pool_out; a.pool_stats = reinterpret_cast<double2*>(pstats); a.pool_mode = 2; a.floor = 0.f;
    return rf_split_zc_launch(a, (hipStream_t)stream);
}
'''
MF = "auto mf = [](const h8& x, const h8& y, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c, 0, 0, 0); };"
NO_MF = "auto mf = [](const h8& x, const h8& y, const f32x4& c) { f32x4 r = c; r[0] += (float)x[0] * (float)y[0]; return r; };"
LOADS = """            st.ph[r] = *reinterpret_cast<const h8*>(p + off);
            st.pl[r] = *reinterpret_cast<const h8*>(p + VOL * 16 + off);"""
NO_LOADS = """            st.ph[r] = h8{(_Float16)(float)off, 0, 0, 0, 0, 0, 0, 0};
            st.pl[r] = h8{(_Float16)(float)r, 0, 0, 0, 0, 0, 0, 0};"""
TILE = "        if (want_pool && kq < 2) {\n#pragma unroll\n            for (int zp = 0; zp < 2; ++zp) {\n                float* t = tile"
NO_TILE = "        if (want_pool && kq < 2 && poolA[0].x == 123.456f) {\n#pragma unroll\n            for (int zp = 0; zp < 2; ++zp) {\n                float* t = tile"
OUTS = "        if (want_pool) {\n            const int co = tid >> 5, pz = (tid >> 3) & 3, w4 = tid & 7;"
NO_OUTS = "        if (want_pool && poolB[0].x == 123.456f) {\n            const int co = tid >> 5, pz = (tid >> 3) & 3, w4 = tid & 7;"
SSTORE = "        stage_store(st, bufB);\n        lds_barrier();                                              // 1:"
NO_SSTORE = "        if (pair < 0) stage_store(st, bufB);\n        lds_barrier();                                              // 1:"
SSTORE2 = "        stage_store(st, bufB);                                      // the next pair's box A"
NO_SSTORE2 = "        if (pair < 0) stage_store(st, bufB);                                      // the next pair's box A"
BAR3 = "        lds_barrier();                                              // 3:"
BAR2 = "        lds_barrier();                                              // 2:"
L2 = "        const unsigned char* p = spre + (size_t)n0 * (2 * VOL * 16);"
L2_HIT = "        const unsigned char* p = spre + (size_t)(n0 & 255) * (2 * VOL * 16);"
W1 = "        stage_store(st, bufB);\n        lds_barrier();                                              // 1:"
W1_STAMP = "        { const long long t0 = __builtin_readcyclecounter(); asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); dbg_wait += __builtin_readcyclecounter() - t0; }\n        stage_store(st, bufB);\n        lds_barrier();                                              // 1:"
W2 = "        stage_store(st, bufB);                                      // the next pair's box A"
W2_STAMP = "        { const long long t0 = __builtin_readcyclecounter(); asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); dbg_wait += __builtin_readcyclecounter() - t0; }\n        stage_store(st, bufB);                                      // the next pair's box A"
DECL = "    ZcStage st;\n    stage_load(st, n_first * 8);"
DECL_STAMP = "    long long dbg_wait = 0; const long long dbg_t0 = __builtin_readcyclecounter();\n    ZcStage st;\n    stage_load(st, n_first * 8);"
END = "        unsigned char* t = bufA; bufA = bufB; bufB = t;\n    }\n}"
END_STAMP = "        unsigned char* t = bufA; bufA = bufB; bufB = t;\n    }\n    if (lane == 0) { a.out[(blockIdx.x * 8 + wave) * 2] = (float)dbg_wait; a.out[(blockIdx.x * 8 + wave) * 2 + 1] = (float)(__builtin_readcyclecounter() - dbg_t0); }\n}"
VARIANTS = {'base': [], 'l2_hit': [(L2, L2_HIT)], 'stamps': [(W1, W1_STAMP), (W2, W2_STAMP), (DECL, DECL_STAMP), (END, END_STAMP)], 'no_mfma': [(MF, NO_MF)], 'no_loads': [(LOADS, NO_LOADS)], 'no_epilogue': [(TILE, NO_TILE), (OUTS, NO_OUTS)],
            'no_loads_no_epilogue': [(LOADS, NO_LOADS), (TILE, NO_TILE), (OUTS, NO_OUTS)],
            'no_staging': [(LOADS, NO_LOADS), (SSTORE, NO_SSTORE), (SSTORE2, NO_SSTORE2)],
            'compute_only': [(LOADS, NO_LOADS), (SSTORE, NO_SSTORE), (SSTORE2, NO_SSTORE2), (TILE, NO_TILE), (OUTS, NO_OUTS)],
            'no_mfma_no_epilogue': [(MF, NO_MF), (TILE, NO_TILE), (OUTS, NO_OUTS)]}


def build(names=None):
    OUT.mkdir(exist_ok=True)
    for name, patches in VARIANTS.items():
        if names and name not in names:
            continue
        src = (CSRC / 'conv3d_split_zc.hip').read_text()
        for old, new in list(patches) + ([(DECL, DECL_STAMP), (END, END_STAMP)] if name != 'stamps' else []):
            assert src.count(old) == 1, (name, old[:60])
            src = src.replace(old, new)
        p = OUT / ('zc_%s.hip' % name)
        p.write_text(src + ENTRY)
        obj = OUT / ('zc_%s.o' % name)
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', str(CSRC), '-c', str(p), '-o', str(obj)], check=True)
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(OUT / ('libzc_%s.so' % name)), str(obj), str(CSRC / 'build' / 'capi.o')], check=True)
        print(name)


def run():
    import torch
    sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
    from rfuse import ops
    dev = torch.device('cuda:0')
    n, cin, edge, cout = 8192, 8, 16, 16
    pre = torch.randint(0, 255, (n * 2 * edge ** 3 * 16,), dtype=torch.uint8, device=dev)
    pre.view(torch.float16).clamp_(-4, 4); pre.view(torch.float16).nan_to_num_(0.0)
    w = ops.pack_conv3_split_weight(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05)
    pooled = torch.empty(n, cout, 8, 8, 8, device=dev); pstats = torch.empty(n, cout, 1, 2, dtype=torch.float64, device=dev)
    VP = ctypes.c_void_p
    names = [a for a in sys.argv[1:]] or list(VARIANTS)
    for rnd in range(2):
        for name in names:
            so = OUT / ('libzc_%s.so' % name)
            if not so.exists():
                continue
            lib = ctypes.CDLL(str(so))
            f = lib.zc_run
            f.argtypes = [VP, VP, VP, VP, ctypes.c_int, VP, VP]
            st = torch.cuda.current_stream().cuda_stream
            dbg = torch.zeros(512 * 8 * 2, device=dev)
            call = lambda: f(pre.data_ptr(), w.data_ptr(), pooled.data_ptr(), pstats.data_ptr(), n, st, dbg.data_ptr())
            for _ in range(5): assert call() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): call()
            e1.record(); torch.cuda.synchronize()
            print('round %d  %-26s %8.1f us' % (rnd, name, e0.elapsed_time(e1) * 50), flush=True)
            if True:
                d = dbg.view(-1, 2)
                print('   waiting for the staging loads: mean %.0f of %.0f cycles per wave (%.1f %%), max %.0f' % (d[:, 0].mean().item(), d[:, 1].mean().item(), 100 * d[:, 0].mean().item() / d[:, 1].mean().item(), d[:, 0].max().item()))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        build(sys.argv[2:])
    else:
        run()
