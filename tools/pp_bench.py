"""Dev tool (GPU box): the dominant launch alone -- rf_conv3d_up_split_presplit, 32+64 -> 56 @8^3 x 8192 (reference model/refinement.py:64-73, dec1 conv1) --
HIP events, whichever kernel the library in RFUSE_LIB dispatches (k_conv3_up_split_pp, or k_conv3_up_split<4> when built with -DRF_UP_PP=0), and the
maximum difference of the pre-split bytes from a reference library's (RFUSE_REF_OUT: a .pt file written by a previous run with --save).
  python tools/pp_bench.py [n=8192] [--pm] [--save file] [--cmp file]      (--pm: the parity-major entry point)"""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops

dev = torch.device('cuda:0')
args = [a for a in sys.argv[1:] if not a.startswith('--')]
n = int(args[0]) if args else 8192
c0, c1, cout, groups = 32, 64, 56, 8
g = torch.Generator().manual_seed(5)
s0 = torch.randn(n, c0, 8, 8, 8, generator=g).relu_().to(dev)
s1 = torch.randn(n, c1, 4, 4, 4, generator=g).relu_().to(dev)
w = (torch.randn(cout, c0 + c1, 3, 3, 3, generator=g) * 0.05).to(dev)
gam, bet = (1 + 0.2 * torch.randn(cout, generator=g)).to(dev), (0.2 * torch.randn(cout, generator=g)).to(dev)
aff = torch.zeros(n, c0 + c1, 4, device=dev)
aff[..., 0] = (0.4 + 0.1 * torch.rand(n, c0 + c1, generator=g)).to(dev)
aff[..., 1] = (1.0 + 0.5 * torch.rand(n, c0 + c1, generator=g)).to(dev)
aff[..., 2] = 0.1
wp = ops.pack_conv3_up_split_weight(w, c0)
PM = '--pm' in sys.argv
fn = lambda: ops.conv3d_up_split_presplit(s0, s1, aff, wp, cout, gam, bet, groups, 1e-5, parity_major=PM)
for _ in range(3):
    out = fn()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 100)
flop = 2 * 27 * (c0 + c1) * cout * 512 * n
print('n=%d  us per launch: %s   min %.1f us = %.3f of 2.5 PF (algorithmic)' % (n, ' '.join('%.1f' % t for t in ts), min(ts), flop / (min(ts) * 1e-6) / 2.5e15))
if '--save' in sys.argv:
    torch.save(out.cpu(), sys.argv[sys.argv.index('--save') + 1])
if '--cmp' in sys.argv:
    ref = torch.load(sys.argv[sys.argv.index('--cmp') + 1])
    a16, b16 = out.cpu().view(torch.float16).float(), ref.view(torch.float16).float()
    # [n][cout/8][h | l][512][8]: value = (h + l / 2048) * 16
    a5, b5 = a16.view(n, cout // 8, 2, 512, 8), b16.view(n, cout // 8, 2, 512, 8)
    if PM:      # slot ((z & 1) 4 + (y & 1) 2 + (x & 1)) 64 + (z >> 1) 16 + (y >> 1) 4 + (x >> 1) -> linear z 64 + y 8 + x
        z, y, x = torch.meshgrid(torch.arange(8), torch.arange(8), torch.arange(8), indexing='ij')
        perm = (((z & 1) * 4 + (y & 1) * 2 + (x & 1)) * 64 + (z >> 1) * 16 + (y >> 1) * 4 + (x >> 1)).reshape(-1)
        a5 = a5[:, :, :, perm]
    va, vb = (a5[:, :, 0] + a5[:, :, 1] / 2048) * 16, (b5[:, :, 0] + b5[:, :, 1] / 2048) * 16
    print('max |this - ref| = %.3e of max |ref| %.3f; bytes equal: %s' % ((va - vb).abs().max().item(), vb.abs().max().item(), torch.equal(out.cpu(), ref)))
