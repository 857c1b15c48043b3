"""Dev probe (GPU box): does the level-0 pair of the retrieval backbone (1 -> 8 -> 16 @16^3 x 8192: k_conv3_cin1_presplit writes a 1.07 GB pre-split
tensor, k_conv3_split_zc reads it) run faster when producer and consumer alternate over sub-batches small enough for the 256 MB Infinity Cache?

    python tools/subbatch_probe.py"""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops
from model.unet import DoubleConv

dev = torch.device('cuda:0')
torch.manual_seed(3)
blk = DoubleConv(1, 16, encoder=True, num_groups=8).to(dev)
c1, c2 = blk.SingleConv1, blk.SingleConv2
g1, g2 = c1.groupnorm, c2.groupnorm
n = 8192
x = torch.randn(n, 1, 16, 16, 16, device=dev)
w1, wp = c1.conv.packed(), c2.conv.packed_split()


def pair(xs):
    m = xs.shape[0]
    pre = ops.conv3d_cin1_presplit(xs, g1.weight, g1.bias, g1.eps, w1, 8, g2.weight, g2.bias, g2.num_groups, g2.eps)
    return ops.conv3d_split_pre_relu(pre, 8, m, 16, wp, 16, pool='only')


for parts in (1, 2, 4, 8, 1, 4):
    m = n // parts
    run = lambda: [pair(x[i * m:(i + 1) * m]) for i in range(parts)]
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    print('%d sub-batches of %d samples: %.3f ms for the pair over all %d samples' % (parts, m, e0.elapsed_time(e1) / 10, n))
