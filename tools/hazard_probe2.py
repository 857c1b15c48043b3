"""Dev tool (GPU box): the two-stream hazard of DESIGN 4.7 taken apart.  Victim = the fp32 box conv (32->32 @16^3 x 8: 128-voxel tiles), built in
several one-patch VARIANTS by tools/hazard_variants.py (run that first, in the CPU container); aggressor = one long-running F16-MFMA workgroup per CU.
For every variant: how many of N launches differ from the solo bits, and WHERE (x inside the tile, lane group / accumulator row).  The `dump`
variant also writes the GroupNorm-applied halo rows it commits to LDS into a debug buffer: are the A operands already different?
    python tools/hazard_variants.py && gpurun -- python tools/hazard_probe2.py"""
import ctypes
import sys
from collections import Counter
from pathlib import Path
import numpy as np
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops
HAZ = REPO / 'tools' / '_haz'
dev = torch.device('cuda:0')
aggr = ctypes.CDLL(str(HAZ / 'aggr.so'))
aggr.launch_aggressor.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
VP = ctypes.c_void_p
CONV_ARGS = [VP, ctypes.c_int, VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, VP, VP, ctypes.c_int, VP, VP]

torch.manual_seed(0)
N, C, E = 8, 32, 16
xs = torch.randn(N, C, E, E, E, device=dev).relu_()
w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
aff = ops.gn_affine(xs, None, torch.ones(C, device=dev), torch.zeros(C, device=dev), 8)
wp = ops.pack_conv3_weight(w)
side = torch.cuda.Stream(dev)
aout = torch.empty(256 * 8 * 256, device=dev)
torch.cuda.synchronize()


def load_variant(name):
    lib = ctypes.CDLL(str(HAZ / ('libv_%s.so' % name)))
    lib.rf_conv3d_k3_gn_relu.argtypes = CONV_ARGS
    lib.rf_conv3d_k3_gn_relu.restype = ctypes.c_int
    return lib


def conv(lib, stream):
    out = torch.empty(N, C, E, E, E, device=dev)
    rc = lib.rf_conv3d_k3_gn_relu(xs.data_ptr(), C, None, 0, N, E, aff.data_ptr(), wp.data_ptr(), C, out.data_ptr(), stream.cuda_stream)
    assert rc == 0, rc
    return out


def fingerprint(v, ref, cnt_x, cnt_gr):
    idx = (v != ref).nonzero().cpu().numpy()
    for (n, co, z, y, x) in idx:
        xin, yin = x & 7, y & 3
        cnt_x[int(xin)] += 1
        cnt_gr[(int((yin & 1) * 2 + (xin >> 2)), int(xin & 3))] += 1
    return len(idx)


def aggress(main):
    rc = aggr.launch_aggressor(0, 1, 256, 60000, aout.data_ptr(), main.cuda_stream)          # one 154-VGPR F16-MFMA workgroup per CU, ~15 ms
    assert rc == 0, rc


main = torch.cuda.current_stream()
ref0 = None
for name in ('control', 'schedbar', 'nops', 'nodma', 'nopfx', 'noslp'):
    lib = load_variant(name)
    ref = conv(lib, main).clone()
    torch.cuda.synchronize()
    if ref0 is None:
        ref0 = ref
    same_as_control = torch.equal(ref, ref0)
    solo_bad = sum(0 if torch.equal(conv(lib, main), ref) else 1 for _ in range(30))
    bad, cnt_x, cnt_gr, nel, maxulp = 0, Counter(), Counter(), 0, 0.0
    for rnd in range(8):
        outs = []
        side.wait_stream(main)
        aggress(main)
        with torch.cuda.stream(side):
            for _ in range(30):
                outs.append(conv(lib, side))
        torch.cuda.synchronize()
        for v in outs:
            if not torch.equal(v, ref):
                bad += 1
                nel += fingerprint(v, ref, cnt_x, cnt_gr)
                d = ((v - ref).abs() / ref.abs().clamp_min(1e-30))
                maxulp = max(maxulp, float(d[v != ref].max()) / 1.19e-7)
    print(f'{name:9s}: solo == control solo: {same_as_control} | solo reruns differing {solo_bad}/30 | beside the F16-MFMA aggressor: {bad}/240 launches differ'
          f' ({nel} elements, max {maxulp:.1f} ulp-ish) | x in tile: {dict(sorted(cnt_x.items()))} | (lane group, acc row): {dict(sorted(cnt_gr.items()))}', flush=True)

# ---- the dump variant: one victim launch per round beside the aggressor; compare the committed halo rows with the solo run's
lib = load_variant('dump')
lib.rf_dbg_set.argtypes = [VP]
BLK, CH, RW = 256, 16, 256                                           # blocks, chunk slots, row slots (x 16 floats)
dbg = torch.zeros(BLK * CH * RW * 16, device=dev)
assert lib.rf_dbg_set(dbg.data_ptr()) == 0
ref = conv(lib, main).clone()
torch.cuda.synchronize()
dref = dbg.clone()
print('dump variant: solo == control solo:', torch.equal(ref, ref0), '| debug floats written:', int((dref != 0).sum()), flush=True)
events = 0
for rnd in range(120):
    dbg.zero_()
    side.wait_stream(main)
    aggress(main)
    with torch.cuda.stream(side):
        v = conv(lib, side)
    torch.cuda.synchronize()
    out_bad = not torch.equal(v, ref)
    dd = (dbg != dref).nonzero().flatten().cpu().numpy()
    if out_bad or len(dd):
        events += 1
        if events <= 12:
            cx, cg = Counter(), Counter()
            nel = fingerprint(v, ref, cx, cg)
            where = Counter()
            mx = 0.0
            for f in dd:
                j = f & 15; row = (f >> 4) % RW; ch = (f >> 4) // RW % CH; blk = (f >> 4) // RW // CH
                where[(int(ch), int(j))] += 1
            if len(dd):
                a, b = dbg[dd].cpu().numpy().astype(np.float64), dref[dd].cpu().numpy().astype(np.float64)
                mx = float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)) / 1.19e-7)
                blks = sorted(set(int((f >> 4) // RW // CH) for f in dd))
                rows = sorted(set(int((f >> 4) % RW) for f in dd))
            print(f'  round {rnd}: output elements differing {nel} (x in tile {dict(sorted(cx.items()))}) | committed halo floats differing {len(dd)}'
                  + (f', max {mx:.1f} ulp-ish, (chunk, halo index) histogram {dict(sorted(where.items()))}, blocks {blks[:12]}, rows {rows[:24]}' if len(dd) else ''), flush=True)
            if len(dd) and events <= 3:
                for f in dd[:6]:
                    print('     e.g. float', int(f), 'got', repr(float(dbg[f])), 'solo', repr(float(dref[f])))
print(f'dump variant: {events}/120 single launches beside the aggressor differed (output or committed rows)', flush=True)
