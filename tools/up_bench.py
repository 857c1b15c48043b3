"""Dev tool (GPU box): time the decoder conv -- fp32-MFMA form (rf_conv3d_up_k3_gn_relu) and split-operand form
(rf_conv3d_up_split_k3_gn_relu) -- on chosen (c0, c1) splits, HIP events."""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
CASES = [('dec1 32+64->56 @8', 256 * B, 32, 64, 8, 56), ('B only 0+64->56 @8', 256 * B, 0, 64, 8, 56), ('A only+1 32+8->56 @8', 256 * B, 32, 8, 8, 56),
         ('C5 dec1 24+48->42 @8', 256 * B, 24, 48, 8, 42), ('dec0 64+128->64 @4', 256 * B, 64, 128, 4, 64), ('final 0+16->16 @64', B, 0, 16, 64, 16), ('C5 unet dec 48+96->78 @32', B // 2, 48, 96, 32, 78), ('C5 unet dec 24+48->24 @32', B // 2, 24, 48, 32, 24)]


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


print('%-24s %10s %8s | %10s %8s %8s' % ('layer', 'fp32 us', 'TF/s', 'split us', 'TF/s eq', 'f16 pipe'))
for name, n, c0, c1, edge, cout in CASES:
    s0 = torch.rand(n, c0, edge, edge, edge, device=dev) if c0 else None
    s1 = torch.rand(n, c1, edge // 2, edge // 2, edge // 2, device=dev)
    cin = c0 + c1
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    aff = torch.zeros(n, cin, 4, device=dev)
    aff[..., 1] = 1.0
    flops = 2 * (27 * c0 + 8 * c1) * cout * edge ** 3 * n
    wp = ops.pack_conv3_up_weight(w, c0)
    us = timed(lambda: ops.conv3d_up_gn_relu(s0, s1, aff, wp, cout))
    line = '%-24s %10.1f %8.1f' % (name, us, flops / us / 1e6)
    if ops.conv_up_split_supported(s0, s1, cout):
        ws = ops.pack_conv3_up_split_weight(w, c0)
        us2 = timed(lambda: ops.conv3d_up_split_gn_relu(s0, s1, aff, ws, cout))
        line += ' | %10.1f %8.1f %7.1f%%' % (us2, flops / us2 / 1e6, 100 * ops.conv_up_split_issued_flops(c0, c1, n, edge, cout) / us2 / 1e6 / 2516)
    print(line)
