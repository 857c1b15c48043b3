"""Times the valid-conv forms of the patch encoders' layers (rf_conv3d_valid_leaky_{valu,lds,mfma,split}) on one layer shape.
usage: python tools/convv_bench.py [n cin s cout k stride] ...   (default: the large layers of PCPatch48 at 1024 windows)"""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'retrieval-fuse_amd'))
import torch
from rfuse import ops

DEV = 'cuda:0'


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    a = [int(v) for v in sys.argv[1:]]
    specs = [tuple(a[i:i + 6]) for i in range(0, len(a), 6)] or [(1024, 12, 44, 24, 3, 1), (1024, 24, 42, 48, 3, 2), (1024, 48, 20, 48, 3, 2)]
    for n, cin, s, cout, k, stride in specs:
        g = torch.Generator().manual_seed(1)
        x = torch.randn(n, cin, s, s, s, generator=g).to(DEV)
        w = (torch.randn(cout, cin, k, k, k, generator=g) / (cin * k ** 3) ** 0.5).to(DEV)
        b = torch.randn(cout, generator=g).to(DEV)
        so = (s - k) // stride + 1
        flops = 2.0 * n * so ** 3 * cout * cin * k ** 3
        line = f'{cin}->{cout} k{k} s{stride} @{s}^3 x{n}: {flops / 1e9:.0f} GFLOP'
        if ops.conv_valid_valu_supported(x, cout, k, stride):
            wt = ops.pack_convv_valu_weight(w)
            t = timed(lambda: ops.conv3d_valid_leaky_valu(x, wt, b, stride, 0.2))
            line += f' | valu {t:.3f} ms {flops / t / 1e9:.0f} TF/s'
        if ops.conv_valid_lds_supported(x, cout, k, stride):
            wl = ops.pack_convv_lds_weight(w)
            t = timed(lambda: ops.conv3d_valid_leaky_lds(x, wl, b, cout, k, stride, 0.2))
            line += f' | lds {t:.3f} ms {flops / t / 1e9:.0f} TF/s'
        if ops.conv_valid_split_supported(x, cout, k, stride):
            ws = ops.pack_convv_split_weight(w, s, stride)
            t = timed(lambda: ops.conv3d_valid_leaky_split(x, ws, b, cout, k, stride, 0.2))
            line += f' | split {t:.3f} ms {flops / t / 1e9:.0f} TF/s (fp32-equivalent)'
        print(line, flush=True)


if __name__ == '__main__':
    main()
