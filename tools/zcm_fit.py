"""Dev tool (GPU box): cycles of the persistent z-column kernel per box and per chunk (pre-split input, full output, no statistics) from layers of 2, 4 and 7
chunks.  Needs tools/_haz/libzcm_base.so (python tools/zcm_ablation.py build base)."""
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
n, edge = 8192, 8
st = torch.cuda.current_stream().cuda_stream
pw = torch.zeros(16, device=dev)
for cin, cout in ((16, 16), (32, 16), (56, 16), (16, 32)):
    src = torch.randint(0, 255, (n * (cin // 8) * 2 * edge ** 3 * 16,), dtype=torch.uint8, device=dev)
    src.view(torch.float16).clamp_(-4, 4); src.view(torch.float16).nan_to_num_(0.0)
    aff = torch.zeros(n, cin, 4, device=dev); aff[..., 1] = 1.0
    w = ops.pack_conv3_split_weight(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05)
    out = torch.empty(n, cout, edge, edge, edge, device=dev)
    call = lambda: f(src.data_ptr(), 1, aff.data_ptr(), w.data_ptr(), out.data_ptr(), None, pw.data_ptr(), pw.data_ptr(), cin, n, edge, st, cout, None, None, None)
    for _ in range(5): assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    cyc = dbg[:4096][dbg[:4096] > 0].mean().item()
    boxes = n * (cout // 16) / 512
    print('%2d -> %2d: %7.1f us  %6.0f k cycles per wave = %5.1f k per box of %d chunks (%.2f k per chunk; MFMA floor 5.4 k per chunk with four waves on a SIMD)' % (
        cin, cout, e0.elapsed_time(e1) * 50, cyc / 1e3, cyc / 1e3 / boxes, cin // 8, cyc / 1e3 / boxes / (cin // 8)))
