"""Dev tool (GPU box): time the parity-split decoder conv (rf_conv3d_up_k3_gn_relu) on chosen (c0, c1) splits, HIP events."""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
CASES = [('dec1 32+64->56 @8', 256 * B, 32, 64, 8, 56), ('B only 0+64->56 @8', 256 * B, 0, 64, 8, 56), ('A + 1 chunk 32+8->56 @8', 256 * B, 32, 8, 8, 56),
         ('1 chunk 0+8->56 @8', 256 * B, 0, 8, 8, 56), ('dec0 64+128->64 @4', 256 * B, 64, 128, 4, 64), ('B only 0+128->64 @4', 256 * B, 0, 128, 4, 64),
         ('final 0+16->16 @64', B, 0, 16, 64, 16)]
print('%-28s %9s %9s' % ('layer', 'us', 'TFLOP/s (executed)'))
for name, n, c0, c1, edge, cout in CASES:
    s0 = torch.rand(n, c0, edge, edge, edge, device=dev) if c0 else None
    s1 = torch.rand(n, c1, edge // 2, edge // 2, edge // 2, device=dev)
    cin = c0 + c1
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    wp = ops.pack_conv3_up_weight(w, c0)
    aff = torch.zeros(n, cin, 4, device=dev); aff[..., 1] = 1.0
    for _ in range(3):
        ops.conv3d_up_gn_relu(s0, s1, aff, wp, cout)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        ops.conv3d_up_gn_relu(s0, s1, aff, wp, cout)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    flops = 2 * (27 * c0 + 8 * c1) * cout * edge ** 3 * n
    print('%-28s %9.1f %9.1f' % (name, us, flops / us / 1e6))
