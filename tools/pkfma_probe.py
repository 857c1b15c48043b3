"""Dev tool (GPU box): are the packed-fp32 VALU instructions bit-stable beside another wave's F16 MFMAs?  (tools/micro/pkfma_probe.hip)
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/_haz/pkfma.so tools/micro/pkfma_probe.hip
    gpurun -- python tools/pkfma_probe.py"""
import ctypes
import itertools
from pathlib import Path
import numpy as np
import torch
HAZ = Path(__file__).resolve().parent / '_haz'
dev = torch.device('cuda:0')
pk = ctypes.CDLL(str(HAZ / 'pkfma.so'))
VP = ctypes.c_void_p
pk.launch_pk.argtypes = [ctypes.c_int, ctypes.c_int, VP, VP, VP, VP, VP, ctypes.c_int, VP]
aggr = ctypes.CDLL(str(HAZ / 'aggr.so'))
aggr.launch_aggressor.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, VP, VP]
NV, BLOCKS, REPS = 8, 1024, 1000
rng = np.random.default_rng(0)
a = rng.standard_normal((NV, 64, 2)).astype(np.float32)
b = (1.7 + 0.3 * rng.standard_normal((NV, 64, 2))).astype(np.float32)
c = np.where(rng.random((NV, 64, 2)) < 0.5, -(a * b) * (1 + 1e-3 * rng.standard_normal((NV, 64, 2))), rng.standard_normal((NV, 64, 2))).astype(np.float32)
inp = torch.from_numpy(np.stack([a, b, c])).to(dev)
res = torch.zeros(BLOCKS * 4, NV, 64, 2, device=dev)
mism = torch.zeros(BLOCKS * 4, NV, 64, dtype=torch.int32, device=dev)
bad = torch.zeros(BLOCKS * 4, NV, 64, 2, device=dev)
aout = torch.empty(256 * 256, device=dev)
main, side = torch.cuda.current_stream(), torch.cuda.Stream(dev)

# kind -> (name, op, sources, op_sel, op_sel_hi): the low result half takes half op_sel[i] of source i, the high half takes half op_sel_hi[i]
bc = np.stack([b[..., 0], c[..., 0]], axis=-1)                      # KIND 1's shared (scale, shift) register pair
KINDS = {
    0: ('v_pk_fma_f32', 'fma', (a, b, c), (0, 0, 0), (1, 1, 1)),
    1: ('v_pk_fma_f32 d, a, s, s op_sel:[0,0,1] op_sel_hi:[1,0,1]', 'fma', (a, bc, bc), (0, 0, 1), (1, 0, 1)),
    2: ('v_pk_mul_f32', 'mul', (a, b), (0, 0), (1, 1)),
    3: ('v_pk_add_f32', 'add', (a, c), (0, 0), (1, 1)),
    4: ('v_fma_f32 x2 (control)', 'fma', (a, b, c), (0, 0, 0), (1, 1, 1)),
    5: ('v_pk_fma_f32 behind own fp32 MFMAs', 'fma', (a, b, c), (0, 0, 0), (1, 1, 1)),
    6: ('v_pk_fma_f32 op_sel_hi:[1,0,1] (src1 low half broadcast)', 'fma', (a, b, c), (0, 0, 0), (1, 0, 1)),
    7: ('v_pk_fma_f32 op_sel:[0,0,1] (src2 high half broadcast)', 'fma', (a, b, c), (0, 0, 1), (1, 1, 1)),
    8: ('v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,0,1], three registers', 'fma', (a, b, c), (0, 0, 1), (1, 0, 1)),
    9: ('v_pk_fma_f32 op_sel_hi:[0,1,1] (src0 low half broadcast)', 'fma', (a, b, c), (0, 0, 0), (0, 1, 1)),
    10: ('v_pk_mul_f32 op_sel_hi:[1,0]', 'mul', (a, b), (0, 0), (1, 0)),
    11: ('v_pk_add_f32 op_sel_hi:[1,0]', 'add', (a, c), (0, 0), (1, 0)),
    12: ('v_pk_fma_f32 op_sel:[0,1,0] (src1 high half broadcast)', 'fma', (a, b, c), (0, 1, 0), (1, 1, 1)),
    13: ('v_pk_fma_f32 op_sel_hi:[1,1,0] (src2 low half broadcast)', 'fma', (a, b, c), (0, 0, 0), (1, 1, 0)),
    14: ('v_pk_fma_f32 op_sel:[1,0,0] op_sel_hi:[0,1,1] (src0 halves swapped)', 'fma', (a, b, c), (1, 0, 0), (0, 1, 1)),
    15: ('v_pk_fma_f32 op_sel:[1,0,0] (src0 high half broadcast)', 'fma', (a, b, c), (1, 0, 0), (1, 1, 1)),
    16: ('v_pk_fma_f32 d, a, SGPR, c op_sel:[1,0,0]', 'fma', (a, np.broadcast_to(b[:, 0:1, :], b.shape).copy(), c), (1, 0, 0), (1, 1, 1)),
    17: ('v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (src1 halves swapped)', 'add', (a, c), (0, 1), (1, 0)),
    18: ('v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,1,0]', 'fma', (a, b, c), (0, 1, 0), (1, 1, 0)),
    19: ('v_pk_mul_f32 op_sel:[0,1] (src1 high half broadcast)', 'mul', (a, b), (0, 1), (1, 1)),
    20: ('v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1] (src0 halves swapped)', 'add', (a, c), (1, 0), (0, 1)),
}


def evaluate(op, srcs, sel_lo, sel_hi, mode='rne'):
    out = np.empty((NV, 64, 2), dtype=np.float32)
    for h, sel in ((0, sel_lo), (1, sel_hi)):
        v = [s[..., sel[i]].astype(np.float64) for i, s in enumerate(srcs)]
        exact = v[0] * v[1] + v[2] if op == 'fma' else v[0] * v[1] if op == 'mul' else v[0] + v[1]
        r = exact.astype(np.float32)
        if mode != 'rne':                                              # directed roundings of the exact (float64) value
            lo = np.where(r.astype(np.float64) > exact, np.nextafter(r, np.float32(-np.inf)), r)
            hi = np.where(r.astype(np.float64) < exact, np.nextafter(r, np.float32(np.inf)), r)
            r = {'down': lo, 'up': hi, 'rtz': np.where(exact >= 0, lo, hi)}[mode]
        out[..., h] = r
    return out


def launch(kind, expect, stream):
    rc = pk.launch_pk(kind, BLOCKS, inp.data_ptr(), res.data_ptr(), expect.data_ptr() if expect is not None else None, mism.data_ptr(), bad.data_ptr(), REPS, stream.cuda_stream)
    assert rc == 0, rc


import sys
ONLY = [int(x) for x in sys.argv[1:]]
for kind, (name, op, srcs, sel_lo, sel_hi) in KINDS.items():
    if ONLY and kind not in ONLY:
        continue
    launch(kind, None, main)
    torch.cuda.synchronize()
    solo = res.cpu().numpy()
    uniform = bool((solo == solo[0]).all())
    exp = torch.from_numpy(solo[0].copy()).to(dev)
    eq_want = int((solo[0] == evaluate(op, srcs, sel_lo, sel_hi)).sum())
    mism.zero_()
    launch(kind, exp, main)
    torch.cuda.synchronize()
    solo_m = int(mism.sum())
    tot = 0
    quarters = np.zeros(4, dtype=np.int64)
    halves = np.zeros(2, dtype=np.int64)
    explain = {}
    for rnd in range(4):
        mism.zero_(); bad.zero_()
        side.wait_stream(main)
        assert aggr.launch_aggressor(0, 1, 256, 60000, aout.data_ptr(), main.cuda_stream) == 0
        with torch.cuda.stream(side):
            launch(kind, exp, side)
        torch.cuda.synchronize()
        m = mism.cpu().numpy()
        tot += int(m.sum())
        quarters += m.reshape(-1, NV, 4, 16).sum(axis=(0, 1, 3))
        if m.sum():
            bd = bad.cpu().numpy()
            w = np.nonzero(m)
            got, ref = bd[w], solo[0][w[1], w[2]]                        # [events, 2]
            ch = got != ref
            halves += ch.sum(axis=0)
            # candidate explanations of the differing halves: another rounding of the same expression, or other operand halves
            cands = {'round ' + md: evaluate(op, srcs, sel_lo, sel_hi, md) for md in ('down', 'up', 'rtz')}
            n = len(srcs)
            for alt_lo in itertools.product((0, 1), repeat=n):
                for alt_hi in itertools.product((0, 1), repeat=n):
                    if (alt_lo, alt_hi) != (tuple(sel_lo), tuple(sel_hi)):
                        cands['halves lo%s hi%s' % (list(alt_lo), list(alt_hi))] = evaluate(op, srcs, alt_lo, alt_hi)
            for key, val in cands.items():
                hit = int((got[ch] == val[w[1], w[2]][ch]).sum())
                explain[key] = explain.get(key, 0) + hit
            explain['_n'] = explain.get('_n', 0) + int(ch.sum())
            if rnd == 0:
                for e in range(min(3, len(got))):
                    k, l = w[1][e], w[2][e]
                    print('      e.g. lane', int(l), 'got', [repr(float(x)) for x in got[e]], 'solo', [repr(float(x)) for x in ref[e]],
                          'sources', [[repr(float(x)) for x in s[k, l]] for s in srcs])
    best = sorted(((v, k) for k, v in explain.items() if k != '_n'), reverse=True)[:3]
    print(f'{name:70s}: solo waves agree {uniform}, == RNE emulation on {eq_want}/{solo[0].size}, solo mismatches {solo_m} | beside the F16-MFMA aggressor: '
          f'{tot} mismatching (lane, repetition) pairs of {4 * BLOCKS * 4 * NV * 64 * REPS:.2e}, by lane quarter {quarters.tolist()}, differing (low, high) halves {halves.tolist()}'
          + (f' | best explanations of {explain["_n"]} differing halves: {best}' if explain else ''), flush=True)
