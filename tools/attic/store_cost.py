import sys, torch
sys.path[:0] = ['/root/repo/retrieval-fuse_amd']
from rfuse import ops
dev = torch.device('cuda:0')
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n, cin, edge, cout in ((8192, 16, 8, 16), (8192, 56, 8, 16), (8192, 8, 16, 16), (32, 16, 64, 16)):
    x = torch.rand(n, cin, edge, edge, edge, device=dev)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    aff = torch.zeros(n, cin, 4, device=dev); aff[..., 1] = 1.0
    ws = ops.pack_conv3_split_weight(w)
    t_full = timed(lambda: ops.conv3d_split_gn_relu(x, aff, ws, cout))
    t_both = timed(lambda: ops.conv3d_split_gn_relu(x, aff, ws, cout, pool='also'))
    t_pool = timed(lambda: ops.conv3d_split_gn_relu(x, aff, ws, cout, pool='only'))
    outb = n * cout * edge ** 3 * 4 / 1e9
    print(f'{cin}->{cout} @{edge}^3 x{n}: full-res out {t_full:.0f} us | out + pooled {t_both:.0f} us | pooled only {t_pool:.0f} us  (full-res output {outb:.2f} GB)', flush=True)
