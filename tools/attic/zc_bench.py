"""Dev tool (GPU box): the level-0 DoubleConv of the retrieval backbone (1 -> 8 -> 16 @16^3) -- the second conv takes the persistent z-column kernel
(csrc/conv3d_split_zc.hip).  Checks it against float64 torch and the plain route, then times the consumer launch alone (HIP events).

    python tools/zc_bench.py [n]"""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops
from model.unet import DoubleConv

dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
torch.manual_seed(3)
blk = DoubleConv(1, 16, encoder=True, num_groups=8).to(dev)
with torch.no_grad():
    for name, p in blk.named_parameters():
        if 'groupnorm.weight' in name:
            p.copy_(1.0 + 0.3 * torch.randn_like(p))
        elif 'groupnorm.bias' in name:
            p.copy_(0.4 * torch.randn_like(p))
        else:
            p.copy_(0.2 * torch.randn_like(p))
c1, c2 = blk.SingleConv1, blk.SingleConv2
g1, g2 = c1.groupnorm, c2.groupnorm
with torch.no_grad():
    xs = torch.randn(1024, 1, 16, 16, 16, device=dev)
    xd = xs.double()
    y = torch.nn.functional.group_norm(xd, 1, g1.weight.double(), g1.bias.double(), 1e-5)
    y = torch.nn.functional.conv3d(y, c1.conv.weight.double(), padding=1).relu()
    y = torch.nn.functional.group_norm(y, 8, g2.weight.double(), g2.bias.double(), 1e-5)
    ref = torch.nn.functional.conv3d(y, c2.conv.weight.double(), padding=1).relu()
    refp = torch.nn.functional.max_pool3d(ref, 2)
    for pool in (None, 'also', 'only'):
        out = blk(xs, pool=pool) if pool else (blk(xs), None)
        ops.USE_PRESPLIT = False
        plain = blk(xs, pool=pool) if pool else (blk(xs), None)
        ops.USE_PRESPLIT = True
        for nm, o, p_, r in (('full', out[0], plain[0], ref), ('pooled', out[1], plain[1], refp)):
            if o is None:
                continue
            sc = r.abs().max().item()
            line = '%-5s %-6s  vs f64 %.2e (plain route %.2e)  vs plain %.2e' % (pool, nm, (o.double() - r).abs().max().item() / sc, (p_.double() - r).abs().max().item() / sc, (o - p_).abs().max().item() / sc)
            st = getattr(o, '_rf_stats', None)
            if st is not None:
                s64 = torch.stack([r.sum(dim=(2, 3, 4)), (r * r).sum(dim=(2, 3, 4))], dim=-1)
                line += '  stats rel err %.2e (tiles %d)' % (((st[0].sum(dim=2) - s64).abs() / s64.abs().clamp_min(1e-9)).max().item(), st[1])
            print(line)
    x = torch.randn(n, 1, 16, 16, 16, device=dev)
    pre = ops.conv3d_cin1_presplit(x, g1.weight, g1.bias, g1.eps, c1.conv.packed(), 8, g2.weight, g2.bias, g2.num_groups, g2.eps)
    wp = c2.conv.packed_split()
    for pool in ('only', None, 'also'):
        run = lambda: ops.conv3d_split_pre_relu(pre, 8, n, 16, wp, 16, pool=pool)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print('consumer 8->16 @16^3 x %d pool=%-5s %8.1f us   %.3f of the f16 pipe (issued)' % (n, pool, us, ops.conv_split_issued_flops(8, n, 16, 16) / us / 1e6 / 2500))
