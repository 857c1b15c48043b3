"""Dev tool (GPU box): the persistent grid form of the split-operand valid conv (rf_conv3d_valid_leaky_split_pg) beside the tile-per-workgroup kernel
(rf_conv3d_valid_leaky_split_ex, split form in and out) on PCPatch48's 12 -> 24 k3 layer: bit equality of the split-form outputs, then HIP-event times.
usage: python tools/convv_pg_bench.py [n s] ...   (default: 2 70, 1 92, 16 140)
Ablations / phase stamps need the dev build: python tools/build_variant.py pgdev conv_valid_split_pg.hip -DRF_PG_DEV, then
RFUSE_LIB=tools/_haz/libpgdev.so python tools/convv_pg_bench.py --ablate <bits> 16 140   |   ... --ablate 8 --stamps"""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'retrieval-fuse_amd'))
import torch
from rfuse import ops

DEV = 'cuda:0'


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def to_split(x):
    """fp32 [n, c, s, s, s] -> SplitActs: [n][c/4][h | l][s^3] 8-byte slots (4 channels of a voxel as f16; h = f16(x/16), l = f16((x/16 - h) * 2^11))"""
    n, c, s = x.shape[0], x.shape[1], x.shape[2]
    t = (x * (1.0 / 16)).clamp(-65504.0, 65504.0)
    h = t.half()
    l = ((t - h.float()) * 2048.0).half()
    hl = torch.stack([h, l], 0).view(2, n, c // 4, 4, s * s * s).permute(1, 2, 0, 4, 3).contiguous()      # [n][c/4][h|l][voxel][4]
    return ops.SplitActs(hl.view(torch.float32).view(n, c, s, s, s))


def from_split(xs):
    """SplitActs -> fp32 [n, c, s, s, s]: (h + l / 2^11) * 16"""
    n, c, s = xs.shape[0], xs.shape[1], xs.shape[2]
    hl = xs.data.view(torch.float16).view(n, c // 4, 2, s * s * s, 4).float()
    v = (hl[:, :, 0] + hl[:, :, 1] * (1.0 / 2048.0)) * 16.0               # [n][c/4][voxel][4]
    return v.permute(0, 1, 3, 2).reshape(n, c, s, s, s)


def main():
    a = [int(v) for v in sys.argv[1:]]
    specs = [tuple(a[i:i + 2]) for i in range(0, len(a), 2)] or [(2, 70), (1, 92), (16, 140)]
    cin, cout, k = 12, 24, 3
    for n, s in specs:
        g = torch.Generator().manual_seed(s)
        xs = to_split(torch.randn(n, cin, s, s, s, generator=g).to(DEV))
        w = (torch.randn(cout, cin, k, k, k, generator=g) / (cin * k ** 3) ** 0.5).to(DEV)
        b = torch.randn(cout, generator=g).to(DEV)
        assert ops.conv_valid_split_pg_supported((n, cin, s), cout, k, 1), (n, s)
        w1, wp = ops.pack_convv_split_weight(w, s, 1), ops.pack_convv_split_pg_weight(w, s, 1)
        ref = ops.conv3d_valid_leaky_split(xs, w1, b, cout, k, 1, 0.2, out_split=True)
        got = ops.conv3d_valid_leaky_split_pg(xs, wp, b, cout, k, 1, 0.2)
        torch.cuda.synchronize()
        r16, g16 = ref.data.view(torch.int16), got.data.view(torch.int16)
        same = torch.equal(r16, g16)
        nbad = int((r16 != g16).sum())
        fr, fg = from_split(ref), from_split(got)
        dmax = float((fr - fg).abs().max())
        extra = ''
        if n * s ** 3 <= 2 * 91 ** 3:
            x64 = from_split(xs).double().cpu()
            want = torch.nn.functional.leaky_relu(torch.nn.functional.conv3d(x64, w.double().cpu(), b.double().cpu()), 0.2)
            extra = f' | vs float64: tile-per-wg {float((fr.double().cpu() - want).abs().max()):.2e}, persistent {float((fg.double().cpu() - want).abs().max()):.2e}'
        so = s - k + 1
        flops = 2.0 * n * so ** 3 * cout * cin * k ** 3
        t1 = timed(lambda: ops.conv3d_valid_leaky_split(xs, w1, b, cout, k, 1, 0.2, out_split=True))
        t2 = timed(lambda: ops.conv3d_valid_leaky_split_pg(xs, wp, b, cout, k, 1, 0.2))
        print(f'12->24 k3 @{s}^3 x{n}: bit-equal {same} ({nbad} differing halves, max |diff| {dmax:.2e}){extra} | tile-per-wg {t1:.3f} ms | persistent {t2:.3f} ms = {flops / t2 / 1e9:.0f} TF/s algorithmic, '
              f'{3 * flops / t2 / 1e9 / 2500:.3f} of the f16 peak issued', flush=True)


def stamps():
    """--ablate 8 --stamps (dev build): phase borders of every workgroup's 9th round"""
    import ctypes
    import numpy as np
    from rfuse import _lib
    n, s, cin, cout, k = 16, 140, 12, 24, 3
    g = torch.Generator().manual_seed(s)
    xs = to_split(torch.randn(n, cin, s, s, s, generator=g).to(DEV))
    w = (torch.randn(cout, cin, k, k, k, generator=g) / (cin * k ** 3) ** 0.5).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    wp = ops.pack_convv_split_pg_weight(w, s, 1)
    for _ in range(3):
        ops.conv3d_valid_leaky_split_pg(xs, wp, b, cout, k, 1, 0.2)
    torch.cuda.synchronize()
    lib = ctypes.CDLL(os.environ.get('RFUSE_LIB') or os.path.join(os.path.dirname(_lib.__file__), 'librfuse_hip.so'))
    buf = (ctypes.c_ulonglong * (1024 * 2 * 8))()
    assert lib.rft_pg_read_stamps(buf) == 0
    st = np.array(buf, dtype=np.uint64).reshape(1024, 2, 8).astype(np.int64)[:256]
    names = ['k-loop', 'barrier (wait for the other team)', 'tile decode + requests', 'epilogue', 'deposit', 'barrier']
    for t in (0, 1):
        print('team %d: round = %.0f ticks of s_memtime (median over workgroups)' % (t, np.median(st[:, t, 6] - st[:, t, 0])))
        for i in range(6):
            print('   %-36s %8.0f' % (names[i], np.median(st[:, t, i + 1] - st[:, t, i])))


if __name__ == '__main__':
    if '--ablate' in sys.argv:
        import ctypes
        i = sys.argv.index('--ablate')
        ctypes.CDLL(os.environ['RFUSE_LIB']).rft_pg_set_ablate(int(sys.argv[i + 1]))
        del sys.argv[i:i + 2]
    if '--stamps' in sys.argv:
        stamps()
        sys.exit(0)
    main()
