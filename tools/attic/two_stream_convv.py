"""Dev tool (GPU box): (1) is rf_conv3d_valid_leaky_split bit-reproducible run to run, (2) do small fp32 convs on a side stream return their
solo bits while it runs on the main stream?  (same experiment as tools/two_stream_bits.py, with the patch encoders' split-operand layers)"""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
xs = torch.randn(8, 32, 16, 16, 16, device=dev).relu_()
wsml = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
affs = ops.gn_affine(xs, None, torch.ones(32, device=dev), torch.zeros(32, device=dev), 8)
wps = ops.pack_conv3_weight(wsml)
vref = ops.conv3d_gn_relu(xs, None, affs, wps, 32).clone()
side = torch.cuda.Stream(dev)
for n, cin, s, cout, k, stride in [(256, 12, 44, 24, 3, 1), (256, 24, 42, 48, 3, 2), (256, 48, 20, 48, 3, 2)]:
    x = torch.randn(n, cin, s, s, s, device=dev)
    w = torch.randn(cout, cin, k, k, k, device=dev) / (cin * k ** 3) ** 0.5
    b = torch.randn(cout, device=dev)
    wp = ops.pack_convv_split_weight(w, s, stride)
    ref = ops.conv3d_valid_leaky_split(x, wp, b, cout, k, stride, 0.2).clone()
    torch.cuda.synchronize()
    bad_solo = sum(0 if torch.equal(ops.conv3d_valid_leaky_split(x, wp, b, cout, k, stride, 0.2), ref) else 1 for _ in range(10))
    badv = badm = 0
    for it in range(10):
        outs = []
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(30):
                outs.append(ops.conv3d_gn_relu(xs, None, affs, wps, 32))
        for _ in range(3):
            o = ops.conv3d_valid_leaky_split(x, wp, b, cout, k, stride, 0.2)
        torch.cuda.synchronize()
        for v in outs:
            if not torch.equal(v, vref):
                badv += 1
                if badv <= 3:
                    d = (v - vref).abs(); idx = (d > 0).nonzero()
                    print('   wrong: count', len(idx), 'max abs', d.max().item(), 'max rel', (d / vref.abs().clamp_min(1e-30)).max().item(), 'samples', idx[:, 0].unique().tolist(), 'couts', idx[:, 1].unique().tolist()[:8], 'z', idx[:, 2].unique().tolist(), flush=True)
        badm += 0 if torch.equal(o, ref) else 1
    print(f'{cin}->{cout} k{k} s{stride} @{s}^3 x{n}: solo reruns wrong {bad_solo}/10 | with a side stream: victim launches wrong {badv}/300, main wrong {badm}/10', flush=True)
