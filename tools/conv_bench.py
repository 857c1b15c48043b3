"""Dev tool (GPU box): time individual conv layers of the C1 networks through rfuse.ops (HIP events, 20 reps)."""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
# (name, n, c0, c1, edge, cout)
LAYERS = [('rb enc0 1->8 @16', 256 * B, 1, 0, 16, 8), ('rb enc0 8->16 @16', 256 * B, 8, 0, 16, 16), ('rb enc1 16->16 @8', 256 * B, 16, 0, 8, 16),
          ('rb enc1 16->32 @8', 256 * B, 16, 0, 8, 32), ('rb enc2 32->64 @4', 256 * B, 32, 0, 4, 64), ('rb enc3 64->128 @2', 256 * B, 64, 0, 2, 128),
          ('rb dec0 192->64 @4', 256 * B, 64, 128, 4, 64), ('rb dec1 96->56 @8', 256 * B, 32, 64, 8, 56), ('rb dec1 56->16 @8', 256 * B, 56, 0, 8, 16),
          ('dec 16->16 @64', B, 0, 16, 64, 16), ('dec 16->16 @64 (2)', B, 16, 0, 64, 16), ('unet 32->16 @32', B, 0, 32, 32, 16)]
if len(sys.argv) > 2 and sys.argv[2] == 'c5':        # the nf = 12 family (C5's U-Net: channel counts that are not multiples of 8)
    LAYERS = [('c5 unet 6->12 @128', B, 6, 0, 128, 12), ('c5 unet 12->12 @64', B, 12, 0, 64, 12), ('c5 unet 12->24 @64', B, 12, 0, 64, 24), ('c5 unet 24->24 @32', B, 24, 0, 32, 24),
              ('c5 unet 24->48 @32', B, 24, 0, 32, 48), ('c2-like 16->32 @64', B, 16, 0, 64, 32), ('c5 12->12 @8', 256 * B, 12, 0, 8, 12), ('c5 42->12 @8', 256 * B, 42, 0, 8, 12), ('c5 12->12 @64', B, 12, 0, 64, 12),
              ('c5 12->24 @8', 256 * B, 12, 0, 8, 24), ('c5 24->24 @8', 256 * B, 24, 0, 8, 24), ('c5 6->12 @16', 256 * B, 6, 0, 16, 12)]
print('%-22s %9s %9s %9s | %9s %9s %8s' % ('layer', 'us', 'TFLOP/s', 'GB/s', 'split us', 'TF/s eq', 'f16 pipe'))
for name, n, c0, c1, edge, cout in LAYERS:
    s0 = torch.rand(n, c0, edge, edge, edge, device=dev) if c0 else None
    s1 = torch.rand(n, c1, edge // 2, edge // 2, edge // 2, device=dev) if c1 else None
    cin = c0 + c1
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    wp = ops.pack_conv3_weight(w)
    aff = torch.zeros(n, cin, 4, device=dev); aff[..., 1] = 1.0
    for _ in range(3):
        ops.conv3d_gn_relu(s0, s1, aff, wp, cout)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        ops.conv3d_gn_relu(s0, s1, aff, wp, cout)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    flops = 2 * 27 * cin * cout * edge ** 3 * n
    byts = 4 * (n * edge ** 3 * (c0 + cout) + (n * (edge // 2) ** 3 * c1 if c1 else 0))
    line = '%-22s %9.1f %9.1f %9.0f' % (name, us, flops / us / 1e6, byts / us / 1e3)
    if ops.conv_split_supported(s0, s1, cout):
        ws = ops.pack_conv3_split_weight(w)
        for _ in range(3):
            ops.conv3d_split_gn_relu(s0, aff, ws, cout)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            ops.conv3d_split_gn_relu(s0, aff, ws, cout)
        e1.record()
        torch.cuda.synchronize()
        us2 = e0.elapsed_time(e1) * 1e3 / reps
        line += ' | %9.1f %9.1f %7.1f%%' % (us2, flops / us2 / 1e6, 100 * ops.conv_split_issued_flops(cin, n, edge, cout) / us2 / 1e6 / 2516)
    print(line)
