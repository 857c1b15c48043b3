"""Dev tool (GPU box): the fused attention feature encoder (rf_attn_mlp_[split_]rows / _volume) at the bench's shapes, HIP events."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
import torch
from rfuse import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
params = []
for (o_, i_) in ((128, 128), (128, 128), (128, 128), (32, 128)):
    params += [(torch.randn(o_, i_, generator=g) * 0.1).to(dev), (torch.randn(o_, generator=g) * 0.1).to(dev)]
packed = ops.pack_attn_mlp(params)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
B, K = 32, 4
for rows in (B * 4096, B * 4096 * K):
    x = torch.randn(rows, 128, device=dev)
    for arith in ('split', 'fp32'):
        ops.CONV_ARITH = arith
        us = t(lambda: ops.attn_mlp_rows(x, packed))
        fl = 2.0 * rows * (3 * 128 * 128 + 128 * 32)
        print('%8d rows  %-5s %8.1f us  %.0f TFLOP/s useful' % (rows, arith, us, fl / us / 1e6))
ops.CONV_ARITH = 'split'
src = torch.randn(B * K, 16, 32, 32, 32, device=dev).relu_()
us = t(lambda: ops.attn_mlp_volume(src, B, K, 16, 32, 32, packed))
print('volume form (%d x 16 x 32^3): %.1f us' % (B * K, us))
