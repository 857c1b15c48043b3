"""Dev tool (GPU box): the final decoder's first conv (up2 + GN + 3^3 conv 16 -> 16 @64^3, no skip source) alone -- k_conv3_up_split_boxp (persistent) or
k_conv3_up_split_box, whichever the library in RFUSE_LIB dispatches -- HIP events, NCDHW and channel-interleaved output.

    python tools/upbox_bench.py [chunks]"""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops

dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(5)
x = torch.randn(n, 16, 32, 32, 32, device=dev)
w = 0.2 * torch.randn(16, 16, 3, 3, 3, device=dev)
gamma, beta = 1.0 + 0.3 * torch.randn(16, device=dev), 0.4 * torch.randn(16, device=dev)
aff = ops.gn_affine(None, x, gamma, beta, 8, 1e-5)
wp = ops.pack_conv3_up_split_weight(w, 0)
for name, run in (('ncdhw', lambda: ops.conv3d_up_split_gn_relu(None, x, aff, wp, 16)), ('ch8', lambda: ops.conv3d_up_split_gn_relu_ch8(x, aff, wp, 16))):
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
    gb = (x.numel() + n * 16 * 64 ** 3) * 4 / 1e9
    print('%-6s n=%d  %.3f ms  %.0f GB/s algorithmic (%.2f of 8 TB/s)' % (name, n, ms, gb / ms * 1e3, gb / ms * 1e3 / 8000))
