"""Dev tool (CPU, numpy): rounding error of the transform-domain form F(2^3, 3^3) of a 3x3x3 convolution against the direct form, both in the split-operand
arithmetic of the conv kernels (fp32 operands as two f16 pieces x = h + l / 2^11, exact f16 x f16 products, fp32 accumulation rounded once per 32-deep k-step,
hi / lo accumulators combined once), measured against float64 -- the gate VERDICT r4 item 1 asks for before a Winograd kernel is written.

    python tools/winograd_error.py [cin] [cout] [samples]

Layer: GroupNorm'd input (zero mean, unit variance per channel group, ReLU'd predecessor) on whole 8^3 samples with zero padding, Kaiming-uniform weights
(nn.Conv3d default init), the shapes of the retrieval backbone's dec1 skip channels (32 -> 56)."""
import sys
import numpy as np

cin = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 56
samples = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rng = np.random.default_rng(0)
E = 8


def split(x, scale):
    """x (float32/64) * scale -> (h, l) float64 arrays holding f16 values with x*scale ~ h + l / 2^11"""
    v = np.clip(np.asarray(x, dtype=np.float32) * np.float32(scale), -65504, 65504).astype(np.float32)
    h = v.astype(np.float16)
    l = ((v - h.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    return h.astype(np.float64), l.astype(np.float64)


def split_dot(a, b, kstep=32):
    """sum_k a[..., k] * b[..., k] in the kernels' arithmetic: a, b float32 (already scaled 2^-4 / 2^4 by the caller's convention: scale product = 1);
    products exact, fp32 accumulation with one rounding per k-step of 32, separate hi / lo accumulators, out = hi + lo / 2^11 (fp32)"""
    ah, al = split(a, 1.0 / 16)
    bh, bl = split(b, 16.0)
    K = a.shape[-1]
    hi = np.zeros(np.broadcast_shapes(a.shape[:-1], b.shape[:-1]), dtype=np.float32)
    lo = np.zeros_like(hi)
    for k0 in range(0, K, kstep):
        s = slice(k0, k0 + kstep)
        hi = (hi.astype(np.float64) + (ah[..., s] * bh[..., s]).sum(-1)).astype(np.float32)
        lo = (lo.astype(np.float64) + (ah[..., s] * bl[..., s]).sum(-1)).astype(np.float32)
        lo = (lo.astype(np.float64) + (al[..., s] * bh[..., s]).sum(-1)).astype(np.float32)
    return (hi.astype(np.float64) + lo.astype(np.float64) / 2048).astype(np.float32)


# F(2, 3) matrices (Lavin & Gray): Y = A^T [ (G g) . (B^T d) ]
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def t3(M, x, dtype):
    """apply M along the last three axes of x, rounding to `dtype` after every 1-D pass (the transforms run in fp32 on the VALU)"""
    for ax in (-3, -2, -1):
        x = np.moveaxis(np.tensordot(M, np.moveaxis(x, ax, 0), axes=(1, 0)), 0, ax).astype(dtype)
    return x


err = {'direct': [], 'winograd': [], 'direct_f32chain': []}
for s in range(samples):
    x = np.maximum(rng.standard_normal((cin, E, E, E)), 0)
    x = (x - x.mean()) / x.std()                                       # GroupNorm'd (one group: the scale is what matters)
    x = x.astype(np.float32)
    bound = 1.0 / np.sqrt(cin * 27)
    w = rng.uniform(-bound, bound, (cout, cin, 3, 3, 3)).astype(np.float32)
    xp = np.zeros((cin, E + 2, E + 2, E + 2), dtype=np.float32)
    xp[:, 1:-1, 1:-1, 1:-1] = x
    # im2col [voxel][cin * 27]
    cols = np.stack([xp[:, dz:dz + E, dy:dy + E, dx:dx + E] for dz in range(3) for dy in range(3) for dx in range(3)], axis=1)      # [cin, 27, E,E,E]
    cols = cols.reshape(cin * 27, E ** 3).T
    wm = w.reshape(cout, cin * 27)
    ref = cols.astype(np.float64) @ wm.astype(np.float64).T             # [voxel, cout]
    # direct form, split arithmetic: K = 27 * cin in k-steps of 32
    d = split_dot(cols[:, None, :], wm[None, :, :])
    err['direct'].append(d.astype(np.float64) - ref)
    # fp32 sequential chain (v_mfma_f32_16x16x4_f32 rounds after every 4 products): the round-1 kernels
    acc = np.zeros((E ** 3, cout), dtype=np.float32)
    for k0 in range(0, cin * 27, 4):
        acc = (acc.astype(np.float64) + cols[:, k0:k0 + 4].astype(np.float64) @ wm[:, k0:k0 + 4].astype(np.float64).T).astype(np.float32)
    err['direct_f32chain'].append(acc.astype(np.float64) - ref)
    # Winograd F(2^3, 3^3): tiles of 2^3 outputs from 4^3 inputs; input transform in fp32, weight transform in float64 (host, once), GEMM over cin in
    # split arithmetic (one k-step of 32 for cin = 32), output transform in fp32
    T = E // 2
    tiles = np.stack([xp[:, 2 * tz:2 * tz + 4, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4] for tz in range(T) for ty in range(T) for tx in range(T)])   # [tiles, cin, 4,4,4]
    V = t3(BT, tiles.astype(np.float64), np.float32)                   # [tiles, cin, 4,4,4] fp32
    U = t3(G, w.astype(np.float64), np.float64).astype(np.float32)     # [cout, cin, 4,4,4] (rounded to fp32 once, then split like any weight)
    Vk = np.moveaxis(V, 1, -1)                                         # [tiles, 4,4,4, cin]
    Uk = np.moveaxis(U, 1, -1)                                         # [cout, 4,4,4, cin]
    M = split_dot(Vk[:, None], Uk[None])                               # [tiles, cout, 4,4,4]
    Y = t3(AT, M.astype(np.float64), np.float32)                       # [tiles, cout, 2,2,2]
    out = np.zeros((cout, E, E, E), dtype=np.float32)
    i = 0
    for tz in range(T):
        for ty in range(T):
            for tx in range(T):
                out[:, 2 * tz:2 * tz + 2, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = Y[i]
                i += 1
    err['winograd'].append(out.reshape(cout, -1).T.astype(np.float64) - ref)
    scale = np.sqrt((ref ** 2).mean())
print('layer %d -> %d @8^3, %d samples; output rms %.3f' % (cin, cout, samples, scale))
base = None
for name in ('direct_f32chain', 'direct', 'winograd'):
    e = np.concatenate([x.ravel() for x in err[name]])
    rms, mx = np.sqrt((e ** 2).mean()), np.abs(e).max()
    if name == 'direct':
        base = rms
    print('%-18s rms err %.3e (%.2e of the output rms)   max %.3e' % (name, rms, rms / scale, mx))
e_w = np.concatenate([x.ravel() for x in err['winograd']])
print('winograd / direct (split arithmetic) rms error ratio: %.1f' % (np.sqrt((e_w ** 2).mean()) / base))
