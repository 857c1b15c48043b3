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
ops.CONV_ARITH = saved
