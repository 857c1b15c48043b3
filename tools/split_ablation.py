"""Dev tool: where does the single-chunk pre-split box kernel (8->16 @16^3 x 8192, pooled-only) spend its time?  Builds one-patch variants of
csrc/conv3d_split.hip (CPU container: `python tools/split_ablation.py build`) and times them on the GPU box (`python tools/split_ablation.py`)."""
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
"""
NO_MFMA = """    for (int n = 0; n < NB; ++n) { hi[n][0] += (float)ah[0] * (float)bh[n][0]; lo[n][0] += (float)al[0] * (float)bl[n][0]; }
"""
EPI = "    conv_box_epilogue<8, 8, 8, 1, 8, 4, NB, (size_t)CS_LDS_BYTES>(a, acc, reinterpret_cast<float*>(lds), tid, lane, wave, n0, z0, y0, x0, cob, lblock);\n}\n\n// ---------------------------------------------------------------------------------------------------------- 4^3 volumes"
NO_EPI = "    if (acc[0][0][0] == 123.456f) a.pool_out[tid] = acc[1][0][1] + acc[2][0][2] + acc[3][0][3];\n}\n\n// ---------------------------------------------------------------------------------------------------------- 4^3 volumes"
LOADS = """                st.ph[r] = *reinterpret_cast<const h8*>(p + (size_t)voff[r] * 16);
                st.pl[r] = *reinterpret_cast<const h8*>(p + (vol + (size_t)voff[r]) * 16);"""
NO_LOADS = """                st.ph[r] = h8{(_Float16)(float)tid, 0, 0, 0, 0, 0, 0, 0};
                st.pl[r] = h8{(_Float16)(float)r, 0, 0, 0, 0, 0, 0, 0};"""
LOADS_F32 = "                for (int j = 0; j < 8; ++j) st.x[r][j] = s0[(size_t)chan(ca * 8 + j) * vol + voff[r]];"
NO_LOADS_F32 = "                for (int j = 0; j < 8; ++j) st.x[r][j] = (float)(tid + j);"
HOOK = "                if constexpr (!ONE && !PRE) {\n                    const int e = (s - 2) * 4 + m, r = e >> 3, j = e & 7;"
NO_HOOK = "                if constexpr (false) {\n                    const int e = (s - 2) * 4 + m, r = e >> 3, j = e & 7;"
VARIANTS = {'base': [], 'no_loads_f32': [(LOADS_F32, NO_LOADS_F32)], 'no_loads_no_conv': [(LOADS_F32, NO_LOADS_F32), (HOOK, NO_HOOK)],
            'no_loads_no_conv_no_epilogue': [(LOADS_F32, NO_LOADS_F32), (HOOK, NO_HOOK), (EPI, NO_EPI)], 'no_mfma': [(MFMA, NO_MFMA)], 'no_epilogue': [(EPI, NO_EPI)], 'no_loads': [(LOADS, NO_LOADS)],
            'no_mfma_no_epilogue': [(MFMA, NO_MFMA), (EPI, NO_EPI)], 'nothing': [(MFMA, NO_MFMA), (EPI, NO_EPI), (LOADS, NO_LOADS)]}

def build():
    OUT.mkdir(exist_ok=True)
    for name, patches in VARIANTS.items():
        src = (CSRC / 'conv3d_split.hip').read_text()
        for old, new in patches:
            assert src.count(old) == 1, (name, old[:60])
            src = src.replace(old, new)
        p = OUT / ('split_%s.hip' % name)
        p.write_text(src)
        obj = OUT / ('split_%s.o' % name)
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', str(CSRC), '-c', str(p), '-o', str(obj)], check=True)
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(OUT / ('libsplit_%s.so' % name)), str(obj), str(CSRC / 'build' / 'capi.o')], check=True)
        print(name)

def run_multi():
    import torch
    sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
    from rfuse import ops
    dev = torch.device('cuda:0')
    VP = ctypes.c_void_p
    for (n, cin, edge, cout) in ((8192, 56, 8, 16), (32, 16, 64, 16), (8192, 16, 8, 32)):
        x = torch.randn(n, cin, edge, edge, edge, device=dev).relu_()
        aff = torch.zeros(n, cin, 4, device=dev); aff[..., 1] = 1.0
        w = ops.pack_conv3_split_weight(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05)
        out = torch.empty(n, cout, edge, edge, edge, device=dev); stats = torch.empty(n, cout, (edge // 8) ** 3, 2, dtype=torch.float64, device=dev)
        for name in ('base', 'no_loads_f32', 'no_loads_no_conv', 'no_loads_no_conv_no_epilogue', 'no_mfma', 'no_epilogue'):
            lib = ctypes.CDLL(str(OUT / ('libsplit_%s.so' % name)))
            f = lib.rf_conv3d_split_k3_gn_relu
            f.argtypes = [VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, VP, VP, ctypes.c_int, VP, VP, VP, VP, VP]
            st = torch.cuda.current_stream().cuda_stream
            call = lambda: f(x.data_ptr(), cin, n, edge, aff.data_ptr(), w.data_ptr(), cout, out.data_ptr(), stats.data_ptr(), None, None, st)
            for _ in range(3): assert call() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): call()
            e1.record(); torch.cuda.synchronize()
            print('%d->%d @%d^3 x %d  %-30s %8.1f us' % (cin, cout, edge, n, name, e0.elapsed_time(e1) * 100), flush=True)


def run():
    import torch
    sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
    from rfuse import ops
    dev = torch.device('cuda:0')
    n, cin, edge, cout = 8192, 8, 16, 16
    pre = torch.randint(0, 255, (n * 2 * edge ** 3 * 16,), dtype=torch.uint8, device=dev)
    pre.view(torch.float16).clamp_(-4, 4); pre.view(torch.float16).nan_to_num_(0.0)
    w = ops.pack_conv3_split_weight(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05)
    pooled = torch.empty(n, cout, 8, 8, 8, device=dev); pstats = torch.empty(n, cout, 8, 2, dtype=torch.float64, device=dev)
    VP = ctypes.c_void_p
    for name in VARIANTS:
        lib = ctypes.CDLL(str(OUT / ('libsplit_%s.so' % name)))
        f = lib.rf_conv3d_split_pre_k3_relu
        f.argtypes = [VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, VP, ctypes.c_int, VP, VP, VP, VP, VP]
        st = torch.cuda.current_stream().cuda_stream
        call = lambda: f(pre.data_ptr(), cin, n, edge, w.data_ptr(), cout, None, None, pooled.data_ptr(), pstats.data_ptr(), st)
        for _ in range(3): assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        print('%-22s %8.1f us' % (name, e0.elapsed_time(e1) * 100), flush=True)

if __name__ == '__main__':
    build() if len(sys.argv) > 1 and sys.argv[1] == 'build' else (run_multi() if len(sys.argv) > 1 and sys.argv[1] == 'multi' else run())
