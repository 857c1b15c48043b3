"""Dev tool (GPU box): run kernels after filling every VGPR of every SIMD with NaN patterns (tests/testkit: rft_poison_vgprs); a kernel that reads a
register it never wrote changes its bits (or turns NaN)."""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd'), str(REPO / 'tests')]
from rfuse import ops, _lib
import testkit
dev = torch.device('cuda:0')
lib = _lib.load()
torch.manual_seed(0)


def check(name, fn, reps=20):
    ref = fn().clone()
    torch.cuda.synchronize()
    bad = nan = 0
    for _ in range(reps):
        assert testkit.load().rft_poison_vgprs(None) == 0
        v = fn()
        torch.cuda.synchronize()
        if not torch.equal(v, ref):
            bad += 1
            nan += int(torch.isnan(v).sum())
    print(f'{name}: launches with different bits after VGPR poison {bad}/{reps} (NaNs {nan})', flush=True)


for (n, c, e, co) in [(8, 32, 16, 32), (8, 16, 32, 16), (64, 32, 8, 32), (256, 16, 8, 16), (64, 64, 4, 64), (64, 128, 2, 128), (4, 12, 64, 12)]:
    xs = torch.randn(n, c, e, e, e, device=dev).relu_()
    w = torch.randn(co, c, 3, 3, 3, device=dev) * 0.05
    aff = ops.gn_affine(xs, None, torch.ones(c, device=dev), torch.zeros(c, device=dev), 4)
    for arith in ('fp32', 'split'):
        ops.CONV_ARITH = arith
        wp = ops.pack_conv3_weight(w)
        check(f'conv3d_gn_relu[{arith}] {c}->{co} @{e}^3 x{n}', lambda: ops.conv3d_gn_relu(xs, None, aff, wp, co))
ops.CONV_ARITH = 'split'
