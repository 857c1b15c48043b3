"""Dev tool (GPU box): the 16 -> 32 @8^3 pre-split layer on the persistent z-column form with outputs switched off one by one (what does each part of
the epilogue cost?).  Needs tools/_haz/libzcm_base.so (python tools/zcm_ablation.py build base)."""
import ctypes, sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops
dev = torch.device('cuda:0')
VP = ctypes.c_void_p
lib = ctypes.CDLL(str(REPO / 'tools' / '_haz' / 'libzcm_base.so'))
f = lib.zcm_run
f.argtypes = [VP, ctypes.c_int, VP, VP, VP, VP, VP, VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, VP, ctypes.c_int, VP, VP, VP]
dbg = torch.zeros(4096 + 4 * 2048, device=dev)
lib.zcm_dbg.argtypes = [VP]; lib.zcm_dbg(dbg.data_ptr())
n, cin, edge, cout = 8192, 16, 8, 32
src = torch.randint(0, 255, (n * (cin // 8) * 2 * edge ** 3 * 16,), dtype=torch.uint8, device=dev)
src.view(torch.float16).clamp_(-4, 4); src.view(torch.float16).nan_to_num_(0.0)
aff = torch.zeros(n, cin, 4, device=dev); aff[..., 1] = 1.0
w = ops.pack_conv3_split_weight(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05)
out = torch.empty(n, cout, edge, edge, edge, device=dev)
pooled = torch.empty(n, cout, 4, 4, 4, device=dev)
st1 = torch.empty(n, cout, 1, 2, dtype=torch.float64, device=dev); st2 = torch.empty_like(st1)
pw = torch.zeros(16, device=dev)
st = torch.cuda.current_stream().cuda_stream
for label, po, s1, s2 in (('full + pooled + both statistics', pooled, st1, st2), ('full + pooled, no statistics', pooled, None, None), ('full only + its statistics', None, st1, None), ('full only', None, None, None)):
    call = lambda: f(src.data_ptr(), 1, aff.data_ptr(), w.data_ptr(), out.data_ptr(), None, pw.data_ptr(), pw.data_ptr(), cin, n, edge, st, cout,
                     po.data_ptr() if po is not None else None, s1.data_ptr() if s1 is not None else None, s2.data_ptr() if s2 is not None else None)
    for _ in range(5): assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    cyc = dbg[:4096][dbg[:4096] > 0].mean().item()
    print('%-36s %7.1f us  %6.0f k cycles per wave' % (label, e0.elapsed_time(e1) * 50, cyc / 1e3))
