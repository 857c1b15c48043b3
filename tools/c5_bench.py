"""Dev tool (GPU box): time the C5-specific layers in isolation (HIP events): the 128^3 / 64^3 U-Net backbone convs (nf = 12) and
the PCPatch48 / Patch32 valid-conv encoder layers.   python tools/c5_bench.py [B]"""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16

def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

print('--- U-Net backbone convs (GroupNorm + conv3 + ReLU), B = %d chunks' % B)
for name, n, c0, c1, edge, cout in [('1->6 @128', B, 1, 0, 128, 6), ('6->12 @128', B, 6, 0, 128, 12), ('12->12 @64', B, 12, 0, 64, 12), ('12->24 @64', B, 12, 0, 64, 24),
                                    ('24->24 @32', B, 24, 0, 32, 24), ('24->48 @32', B, 24, 0, 32, 48), ('48->96 @16', B, 48, 0, 16, 96), ('288->96 @16', B, 96, 192, 16, 96),
                                    ('144->78 @32', B, 48, 96, 32, 78), ('78->12 @32', B, 78, 0, 32, 12)]:
    s0 = torch.rand(n, c0, edge, edge, edge, device=dev) if c0 else None
    s1 = torch.rand(n, c1, edge // 2, edge // 2, edge // 2, device=dev) if c1 else None
    cin = c0 + c1
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    aff = torch.zeros(n, cin, 4, device=dev); aff[..., 1] = 1.0
    if ops.conv_up_supported(s0, s1, cout):
        wp = ops.pack_conv3_up_weight(w, c0)
        ms = timeit(lambda: ops.conv3d_up_gn_relu(s0, s1, aff, wp, cout))
    else:
        wp = ops.pack_conv3_weight(w)
        ms = timeit(lambda: ops.conv3d_gn_relu(s0, s1, aff, wp, cout))
    flops = 2 * 27 * cin * cout * edge ** 3 * n
    byts = 4 * (n * edge ** 3 * (c0 + cout) + (n * (edge // 2) ** 3 * c1 if c1 else 0))
    print('%-14s %9.3f ms  %7.1f TF/s (direct form)  %7.0f GB/s (in+out once)' % (name, ms, flops / ms / 1e9, byts / ms / 1e6))

print('--- valid-conv patch encoders, %d windows' % (64 * B))
for enc, nf, s0, spec in [('PCPatch48 nf=12', 12, 48, ((0, 1, 5, 1), (1, 2, 3, 1), (2, 4, 3, 2), (4, 4, 3, 2), (4, 8, 3, 2), (8, 8, 3, 1), (8, 8, 2, 1))),
                          ('Patch32 nf=8', 8, 32, ((0, 1, 5, 1), (1, 2, 3, 1), (2, 4, 3, 2), (4, 8, 3, 1), (8, 8, 3, 2), (8, 8, 4, 1))),
                          ('Patch08 nf=16', 16, 8, ((0, 1, 3, 1), (1, 4, 3, 1), (4, 4, 3, 1), (4, 8, 2, 1)))]:
    n, s, tot = 64 * B, s0, 0.0
    for cin_m, cout_m, k, stride in spec:
        cin, cout = (1 if cin_m == 0 else cin_m * nf), cout_m * nf
        x = torch.rand(n, cin, s, s, s, device=dev)
        w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
        b = torch.zeros(cout, device=dev)
        wp = ops.pack_convv_weight(w)
        ms = timeit(lambda: ops.conv3d_valid_leaky_mfma(x, wp, b, cout, k, stride, 0.2))
        so = (s - k) // stride + 1
        flops = 2 * cin * k ** 3 * cout * so ** 3 * n
        ms2 = None
        if ops.conv_valid_lds_supported(x, cout, k, stride):
            wl = ops.pack_convv_lds_weight(w)
            ms2 = timeit(lambda: ops.conv3d_valid_leaky_lds(x, wl, b, cout, k, stride, 0.2))
        ms3 = None
        if ops.conv_valid_valu_supported(x, cout, k, stride):
            wt = ops.pack_convv_valu_weight(w)
            ms3 = timeit(lambda: ops.conv3d_valid_leaky_valu(x, wt, b, stride, 0.2))
        print('%-16s %3d->%-3d k%d s%d  %2d^3->%2d^3  gather %9.3f ms %6.1f TF/s   lds %s   valu %s' % (enc, cin, cout, k, stride, s, so, ms, flops / ms / 1e9,
              '%9.3f ms %6.1f TF/s' % (ms2, flops / ms2 / 1e9) if ms2 else '-', '%9.3f ms %6.1f TF/s' % (ms3, flops / ms3 / 1e9) if ms3 else '-'))
        tot += min(t for t in (ms, ms2, ms3) if t)
        s = so
    print('%-16s total %.3f ms' % (enc, tot))
