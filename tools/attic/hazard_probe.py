"""Dev tool (GPU box): which property of a co-resident kernel moves the bits of small fp32-MFMA convs on another stream?
Synthetic aggressors (tools/micro/hazard_probe.hip): f16 MFMA / fp32 MFMA / VALU only x ~110 / ~160 / ~240 VGPRs.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/hazard_probe.so tools/micro/hazard_probe.hip && python tools/hazard_probe.py"""
import ctypes, sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops
dev = torch.device('cuda:0')
lib = ctypes.CDLL('/tmp/hazard_probe.so')
lib.launch_aggressor.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
torch.manual_seed(0)
xs = torch.randn(8, 32, 16, 16, 16, device=dev).relu_()
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
aff = ops.gn_affine(xs, None, torch.ones(32, device=dev), torch.zeros(32, device=dev), 8)
wp = ops.pack_conv3_weight(w)
saved = ops.CONV_ARITH
ops.CONV_ARITH = 'fp32'
vref = ops.conv3d_gn_relu(xs, None, aff, wp, 32).clone()
side = torch.cuda.Stream(dev)
blocks = 256 * 8
out = torch.empty(blocks * 256, device=dev)
torch.cuda.synchronize()
for kind, kname in ((0, 'f16 MFMA'), (1, 'fp32 MFMA'), (2, 'VALU only')):
    for regs, rname in ((0, '~110 VGPRs, 4 waves/SIMD'), (1, '~160 VGPRs, 3'), (2, '~240 VGPRs, 2')):
        bad = 0
        for it in range(6):
            outs = []
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(30):
                    outs.append(ops.conv3d_gn_relu(xs, None, aff, wp, 32))
            for _ in range(4):
                rc = lib.launch_aggressor(kind, regs, blocks, 3000, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
            torch.cuda.synchronize()
            bad += sum(0 if torch.equal(v, vref) else 1 for v in outs)
        print(f'aggressor {kname:10s} {rname:26s}: victim launches with different bits {bad}/180', flush=True)

bad = 0
for it in range(6):
    outs = []
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(30):
            outs.append(ops.conv3d_gn_relu(xs, None, aff, wp, 32))
    lib.launch_aggressor(0, 1, 256, 60000, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    bad += sum(0 if torch.equal(v, vref) else 1 for v in outs)
print(f'library victim beside ONE f16-MFMA aggressor workgroup per CU (one wave per SIMD): different bits in {bad}/180 launches', flush=True)

# ---- what does a disturbed fp32 MFMA chain compute?  A synthetic victim (one chain of K MFMAs per wave, same operands in every wave)
import numpy as np
lib.launch_victim.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
K, VB = 64, 4096
rng = np.random.default_rng(0)
A = rng.standard_normal((K, 16, 4)).astype(np.float32)             # [k-step][row][kk]
B = rng.standard_normal((K, 4, 16)).astype(np.float32)             # [k-step][kk][col]
lanes = np.arange(64)
a_l = torch.from_numpy(np.ascontiguousarray(A[:, lanes & 15, lanes >> 4])).to(dev)      # lane: row = l & 15, kk = l >> 4
b_l = torch.from_numpy(np.ascontiguousarray(B[:, lanes >> 4, lanes & 15])).to(dev)      # lane: col = l & 15, kk = l >> 4
vout = torch.empty(VB * 64 * 4, device=dev)


def emulate(fused):
    acc = np.zeros((16, 16), dtype=np.float32)
    for k in range(K):
        if fused:                                                   # one rounding per MFMA: acc + exact dot-4
            acc = (acc.astype(np.longdouble) + (A[k].astype(np.longdouble) @ B[k].astype(np.longdouble))).astype(np.float32)
        else:                                                       # four fp32 FMAs in kk order
            for kk in range(4):
                acc = (acc.astype(np.longdouble) + np.outer(A[k][:, kk].astype(np.longdouble), B[k][kk].astype(np.longdouble))).astype(np.float32)
    return acc


def tile_of(flat):                                                  # [64 lanes][4] -> [row][col]: lane: col = l & 15, rows 4 (l >> 4) + r
    t = np.zeros((16, 16), dtype=np.float32)
    v = flat.reshape(64, 4)
    for l in range(64):
        t[4 * (l >> 4):4 * (l >> 4) + 4, l & 15] = v[l]
    return t


lib.launch_victim(a_l.data_ptr(), b_l.data_ptr(), K, VB, vout.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
solo = vout.cpu().numpy().reshape(VB, 256)
assert (solo == solo[0]).all(), 'solo victim waves disagree'
chain, fused = emulate(False), emulate(True)
st = tile_of(solo[0])
print('solo victim == fp32 FMA chain emulation:', bool((st == chain).all()), '| == one-rounding-per-MFMA emulation:', bool((st == fused).all()),
      '| elements where the two emulations differ:', int((chain != fused).sum()), 'of 256', flush=True)
diff_waves = 0
seen = []
for it in range(10):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(20):
            lib.launch_victim(a_l.data_ptr(), b_l.data_ptr(), K, VB, vout.data_ptr(), side.cuda_stream)
    # one aggressor workgroup per CU (one wave per SIMD, 154 VGPRs): the victims certainly share its SIMDs
    lib.launch_aggressor(0, 1, 256, 60000, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = vout.cpu().numpy().reshape(VB, 256)
    bad = np.nonzero((got != solo[0]).any(axis=1))[0]
    diff_waves += len(bad)
    for w in bad[:3]:
        seen.append(tile_of(got[w]))
print(f'beside the f16-MFMA aggressor: {diff_waves} victim waves (of 10 x {VB} checked after the last launch) returned different bits', flush=True)
for t in seen[:4]:
    d = t != st
    print('   a disturbed wave: elements changed', int(d.sum()), '| of those equal to the one-rounding emulation', int((t[d] == fused[d]).sum()),
          '| max |diff| in ulp-ish', float(np.max(np.abs(t[d] - st[d]) / np.maximum(np.abs(st[d]), 1e-30)) / 1.19e-7), flush=True)

# ---- which of the library's fp32 kernels are sensitive?  (one F16-MFMA aggressor workgroup per CU beside 30 launches, 4 rounds)
import torch.nn.functional as F


def sensitive(name, fn, rounds=4):
    ref = fn().clone()
    torch.cuda.synchronize()
    bad = 0
    for _ in range(rounds):
        outs = []
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(30):
                outs.append(fn())
        lib.launch_aggressor(0, 1, 256, 60000, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(v, ref) else 1 for v in outs)
    print(f'{name}: different bits in {bad}/{30 * rounds} launches', flush=True)


g = torch.Generator(device='cpu').manual_seed(1)
def rn(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dev)
# box tiles of the fp32 conv (8^3 boxes, n large)
xb = rn(2048, 16, 8, 8, 8).relu_(); wb = rn(16, 16, 3, 3, 3, scale=0.05)
ab = ops.gn_affine(xb, None, torch.ones(16, device=dev), torch.zeros(16, device=dev), 8); wpb = ops.pack_conv3_weight(wb)
sensitive('k_conv3_mfma, 8^3 box tiles (16->16 @8^3 x 2048)', lambda: ops.conv3d_gn_relu(xb, None, ab, wpb, 16))
# position-major 4^3
x4 = rn(64, 32, 4, 4, 4).relu_(); w4 = rn(32, 32, 3, 3, 3, scale=0.05)
a4 = ops.gn_affine(x4, None, torch.ones(32, device=dev), torch.zeros(32, device=dev), 8); wp4 = ops.pack_conv3_weight(w4)
sensitive('k_conv3_small, position-major 4^3 (32->32 x 64)', lambda: ops.conv3d_gn_relu(x4, None, a4, wp4, 32))
# generic direct (VALU) conv
sensitive('direct VALU conv (32->32 @16^3 x 8)', lambda: ops.conv3d_gn_relu(xs, None, aff, None, 32, direct_weight=w))
# fp32 MFMA valid conv (LDS-staged, no LDS-DMA of weights?) and the VALU valid conv
xv = rn(64, 24, 22, 22, 22); wv = rn(24, 24, 3, 3, 3, scale=0.07); bv = rn(24)
wvl = ops.pack_convv_lds_weight(wv)
sensitive('k_convv_lds (fp32 MFMA valid conv 24->24 s2 @22^3 x 64)', lambda: ops.conv3d_valid_leaky_lds(xv, wvl, bv, 24, 3, 2, 0.2))
xv1 = rn(64, 12, 24, 24, 24); wv1 = rn(24, 12, 3, 3, 3, scale=0.07); wvt = ops.pack_convv_valu_weight(wv1)
sensitive('k_convv_valu (VALU valid conv 12->24 @24^3 x 64)', lambda: ops.conv3d_valid_leaky_valu(xv1, wvt, bv, 1, 0.2))
ops.CONV_ARITH = saved
