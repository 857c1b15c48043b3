import os, sys
os.environ['RFDBG'] = '1'
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops, _lib
from rfuse.ops import _p, _stream
dev = torch.device('cuda:0'); lib = _lib.load()
for name, n, cin, edge, cout in [('56->16@8', 8192, 56, 8, 16), ('24->16@8', 8192, 24, 8, 16)]:
    x = torch.randn(n, cin, edge, edge, edge, device=dev).relu_()
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    aff = ops.gn_affine(x, None, torch.ones(cin, device=dev), torch.zeros(cin, device=dev), 8)
    ws = ops.pack_conv3_split_weight(w)
    nb = n * (edge // 8) ** 3
    out = torch.empty(n, cout, edge, edge, edge, device=dev)
    dbg = torch.zeros(nb * 8, dtype=torch.int64, device=dev)
    for rep in range(2):
        _lib.check(lib.rf_conv3d_split_k3_gn_relu(_p(x), cin, n, edge, _p(aff), _p(ws), cout, _p(out), None, None, _p(dbg), _stream()), 'k')
    torch.cuda.synchronize()
    t = dbg.view(nb, 8).cpu().numpy().astype(np.float64)
    d = np.diff(t[:, :6], axis=1)
    names = ['steps 0-1', 'step 2', 'steps 3-6', 'LDS store of next chunk', 'barrier']
    print(name, 'chunk 2: total mean %.0f' % (t[:, 5] - t[:, 0]).mean())
    for i, nm in enumerate(names):
        print('    %-28s mean %7.0f  p50 %7.0f  p90 %7.0f' % (nm, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
