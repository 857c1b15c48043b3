"""Dev tool: the tile plan of the split valid conv scores tiles by MFMA efficiency only; variants that also charge the staged positions per output voxel
(halo re-reads = conversion work).  `build` (CPU container) makes one library per ALPHA; run on the GPU box to time the big-volume layers."""
import ctypes, subprocess, sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
CSRC = REPO / 'retrieval-fuse_amd' / 'csrc'
OUT = REPO / 'tools' / '_haz'
ALPHAS = ['0.0', '0.02', '0.04', '0.08']
OLD = "                    const double eff = tile_eff * (double)(k3 * cgc) / (8.0 * ksteps);"


def build():
    OUT.mkdir(exist_ok=True)
    for al in ALPHAS:
        src = (CSRC / 'conv_valid_split.hip').read_text()
        assert src.count(OLD) == 1
        src = src.replace(OLD, OLD[:-1] + " * (1.0 - %s * (double)npos / V);" % al)
        p = OUT / ('cvs_%s.hip' % al)
        p.write_text(src)
        obj = OUT / ('cvs_%s.o' % al)
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', str(CSRC), '-c', str(p), '-o', str(obj)], check=True)
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(OUT / ('libcvs_%s.so' % al)), str(obj), str(CSRC / 'build' / 'capi.o')], check=True)
        print(al, flush=True)


def run():
    import torch
    dev = torch.device('cuda:0')
    VP, I = ctypes.c_void_p, ctypes.c_int
    layers = [(16, 12, 140, 24, 3, 1), (16, 24, 138, 48, 3, 2), (16, 48, 68, 48, 3, 2), (32, 8, 76, 16, 3, 1), (32, 16, 74, 32, 3, 2), (32, 32, 36, 64, 3, 1),
              (1024, 12, 44, 24, 3, 1), (1024, 24, 42, 48, 3, 2), (1024, 48, 20, 48, 3, 2)]
    for (n, cin, s, cout, k, st) in layers:
        x = torch.randn(n, cin, s, s, s, device=dev)
        w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
        b = torch.randn(cout, device=dev)
        so = (s - k) // st + 1
        out = torch.empty(n, cout, so, so, so, device=dev)
        line = '%4d x %3d -> %3d @%3d^3 s%d:' % (n, cin, cout, s, st)
        for al in ALPHAS:
            lib = ctypes.CDLL(str(OUT / ('libcvs_%s.so' % al)))
            lib.rf_convv_split_packed_bytes.restype = ctypes.c_size_t
            lib.rf_convv_split_packed_bytes.argtypes = [I, I, I, I, I]
            nbytes = lib.rf_convv_split_packed_bytes(cout, cin, k, s, st)
            if not nbytes:
                line += '   %s: --' % al
                continue
            wp = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            lib.rf_convv_split_pack_weight.argtypes = [VP, I, I, I, I, I, VP, VP]
            stream = torch.cuda.current_stream().cuda_stream
            assert lib.rf_convv_split_pack_weight(w.data_ptr(), cout, cin, k, s, st, wp.data_ptr(), stream) == 0
            f = lib.rf_conv3d_valid_leaky_split
            f.argtypes = [VP, I, I, I, VP, VP, I, I, I, ctypes.c_float, VP, VP]
            call = lambda: f(x.data_ptr(), n, cin, s, wp.data_ptr(), b.data_ptr(), cout, k, st, 0.2, out.data_ptr(), stream)
            for _ in range(2):
                assert call() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call()
            e1.record(); torch.cuda.synchronize()
            line += '   %s: %7.3f ms' % (al, e0.elapsed_time(e1) / 5)
        print(line, flush=True)


if __name__ == '__main__':
    build() if len(sys.argv) > 1 and sys.argv[1] == 'build' else run()
