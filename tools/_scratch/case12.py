import sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd'), str(REPO / 'tests')]
import torch, torch.nn.functional as F
from oracle import refpath
from rfuse import ops
from model.unet import SingleConv
gpu = torch.device('cuda:0')
def rel_err(got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    return float((got - ref).abs().max() / max(1e-12, ref.abs().max()))
for case in [(1030, 8, 8, 8, 40, 4), (1030, 16, 0, 8, 32, 8)]:
    n, c0, c1, edge, cout, groups = case
    gen = torch.Generator().manual_seed(sum(case)); cin = c0 + c1
    torch.manual_seed(100 + sum(case))
    layer = SingleConv(cin, cout, num_groups=groups)
    with torch.no_grad():
        layer.groupnorm.weight.copy_(1 + 0.3 * torch.randn(cin, generator=gen)); layer.groupnorm.bias.copy_(0.3 * torch.randn(cin, generator=gen))
    layer.to(gpu)
    x0 = torch.randn(n, c0, edge, edge, edge, generator=gen).relu() if c0 else None
    x1 = torch.randn(n, c1, edge // 2, edge // 2, edge // 2, generator=gen).relu() if c1 else None
    r = torch.randn(n, cout, edge, edge, edge, generator=gen)
    sd = {'p.groupnorm.weight': layer.groupnorm.weight.detach().cpu().double().requires_grad_(True),
          'p.groupnorm.bias': layer.groupnorm.bias.detach().cpu().double().requires_grad_(True),
          'p.conv.weight': layer.conv.weight.detach().cpu().double().requires_grad_(True)}
    o0 = x0.double().requires_grad_(True) if c0 else None
    o1 = x1.double().requires_grad_(True) if c1 else None
    parts = ([o0] if c0 else []) + ([F.interpolate(o1, scale_factor=2, mode='nearest')] if c1 else [])
    yo = refpath.single_conv_gcr(torch.cat(parts, 1), sd, 'p', groups)
    (yo * r.double()).sum().backward()
    for arith in ('split', 'fp32'):
        ops.CONV_ARITH = arith
        for p_ in layer.parameters(): p_.grad = None
        ins = [t.to(gpu).requires_grad_(True) if t is not None else None for t in (x0, x1)]
        y = layer(ins[0], ins[1])
        (y * r.to(gpu)).sum().backward()
        flips = int(((y.cpu() > 0) != (yo > 0)).sum())
        errs = {'y': rel_err(y, yo), 'dW': rel_err(layer.conv.weight.grad, sd['p.conv.weight'].grad), 'dgamma': rel_err(layer.groupnorm.weight.grad, sd['p.groupnorm.weight'].grad),
                'dbeta': rel_err(layer.groupnorm.bias.grad, sd['p.groupnorm.bias'].grad)}
        if c0: errs['dx0'] = rel_err(ins[0].grad, o0.grad)
        if c1: errs['dx1'] = rel_err(ins[1].grad, o1.grad)
        print(case, arith, 'mask flips', flips, {k: '%.1e' % v for k, v in errs.items()}, flush=True)
