"""Dev tool (GPU box): the 2^3 GEMM form (rf_conv3d_e2_split_k3_gn_relu) against the fp32 position-major kernel, launch by launch (HIP events)."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
import torch
from rfuse import ops
dev = torch.device('cuda:0')
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n, cin, cout in [(8192, 64, 64), (8192, 64, 128), (4096, 48, 96), (16384, 64, 128)]:
    x = torch.rand(n, cin, 2, 2, 2, device=dev)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    aff = torch.zeros(n, cin, 4, device=dev); aff[..., 1] = 1.0
    wp, we = ops.pack_conv3_weight(w), ops.pack_conv3_e2_split_weight(w, 2)
    a = t(lambda: ops.conv3d_e2_split_gn_relu(x, aff, we, cout))
    saved, ops.CONV_ARITH = ops.CONV_ARITH, 'fp32'
    b = t(lambda: ops.conv3d_gn_relu(x, None, aff, wp, cout))
    ops.CONV_ARITH = saved
    fl = 2.0 * n * 8 * cin * 8 * cout
    print('%5d x %3d->%3d @2^3: GEMM form %7.1f us (%.0f TFLOP/s useful, %.2f of the f16 pipe issued)   fp32 position-major %7.1f us' % (n, cin, cout, a, fl / a / 1e6, 3 * fl / a / 1e6 / 2500, b))
