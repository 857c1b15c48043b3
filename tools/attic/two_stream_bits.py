"""Dev tool (GPU box): do small fp32 convs on a side stream return their solo bits while split-operand box convs run on the main stream?
(the experiment behind tests/test_kernels_gpu.py::test_split_box_kernel_leaves_concurrent_kernels_alone; `up` as argument: the victim reads an upsampled source)"""
import sys, os
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops, _lib
dev = torch.device('cuda:0')
lib = _lib.load()
torch.manual_seed(0)
n, cin, edge = 2048, 16, 8
x = torch.randn(n, cin, edge, edge, edge, device=dev).relu_()
aff = ops.gn_affine(x, None, torch.ones(cin, device=dev), torch.zeros(cin, device=dev), 8)
# victim: small fp32 convs (128-voxel tiles), as the U-Net backbone's 16^3 layers at B = 8
xs = torch.randn(8, 32, 16, 16, 16, device=dev).relu_()
wsml = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
affs = ops.gn_affine(xs, None, torch.ones(32, device=dev), torch.zeros(32, device=dev), 8)
wps = ops.pack_conv3_weight(wsml)
vref = ops.conv3d_gn_relu(xs, None, affs, wps, 32).clone()
side = torch.cuda.Stream(dev)
torch.cuda.synchronize()
for cout in (32,):
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    ws = ops.pack_conv3_split_weight(w)
    mref = ops.conv3d_split_gn_relu(x, aff, ws, cout).clone()
    wp32 = ops.pack_conv3_weight(w)
    for kind in ('split',):
        badv = badm = 0
        for it in range(15):
            outs = []
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(30):
                    outs.append(ops.conv3d_gn_relu(xs, None, affs, wps, 32))
            for _ in range(6):
                o = ops.conv3d_split_gn_relu(x, aff, ws, cout) if kind == 'split' else ops.conv3d_gn_relu(x, None, aff, wp32, cout)
            torch.cuda.synchronize()
            for v in outs:
                if not torch.equal(v, vref):
                    badv += 1
                    if badv <= 6:
                        d = (v - vref).abs(); idx = (d > 0).nonzero()
                        print('   wrong: count', len(idx), 'max', d.max().item(), 'nan', int(torch.isnan(v).sum()), 'samples', idx[:,0].unique().tolist(), 'couts', idx[:,1].unique().tolist(),
                              'z', idx[:,2].unique().tolist(), 'y', idx[:,3].unique().tolist(), 'x', idx[:,4].unique().tolist())
                        i0 = idx[0].tolist(); print('      e.g. at', i0, 'got', v[tuple(i0)].item(), 'ref', vref[tuple(i0)].item())
            if kind == 'split':
                badm += 0 if torch.equal(o, mref) else 1
        print('main: %s conv 16->%d @8^3 x 2048 | victim launches wrong: %d of 450, main wrong: %d' % (kind, cout, badv, badm))
