"""Dev tool (GPU box): the 4^3-level layers of the retrieval backbone alone (k_conv3_split_s4, k_conv3_up_split_s4), HIP events.

    python tools/s4_bench.py"""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops

dev = torch.device('cuda:0')
torch.manual_seed(4)
n = 8192


def timeit(run, label, flop):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print('%-28s %.3f ms  %.0f TFLOP/s issued (%.2f of 2500)' % (label, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500))


for cin, cout in ((64, 64), (32, 64), (32, 32)):
    x = torch.randn(n, cin, 4, 4, 4, device=dev).relu_()
    g, b = 1 + 0.2 * torch.randn(cin, device=dev), 0.2 * torch.randn(cin, device=dev)
    aff = ops.gn_affine(x, None, g, b, 8)
    wp = ops.pack_conv3_split_weight(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05)
    timeit(lambda: ops.conv3d_split_gn_relu(x, aff, wp, cout), 'split_s4 %d -> %d' % (cin, cout), ops.conv_split_issued_flops(cin, n, 4, cout))
x0 = torch.randn(n, 64, 4, 4, 4, device=dev).relu_()
x1 = torch.randn(n, 128, 2, 2, 2, device=dev).relu_()
g, b = 1 + 0.2 * torch.randn(192, device=dev), 0.2 * torch.randn(192, device=dev)
aff = ops.gn_affine(x0, x1, g, b, 8)
wp = ops.pack_conv3_up_split_weight(torch.randn(64, 192, 3, 3, 3, device=dev) * 0.05, 64)
timeit(lambda: ops.conv3d_up_split_gn_relu(x0, x1, aff, wp, 64), 'up_split_s4 64+128 -> 64', ops.conv_up_split_issued_flops(64, 128, n, 4, 64))
