"""Dev tool (GPU box): the level-0 pre-split producer (rf_conv3d_cin1_presplit, 1 -> 8 @16^3) on the bench's 8192 patches, HIP events."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
import torch
from rfuse import ops
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
x = torch.rand(n, 1, 16, 16, 16, device=dev)
w = torch.randn(8, 1, 3, 3, 3, device=dev) * 0.2
wp = ops.pack_conv3_weight(w)
g1, b1 = torch.ones(1, device=dev), torch.zeros(1, device=dev)
g8, b8 = torch.ones(8, device=dev), torch.zeros(8, device=dev)
def run(): return ops.conv3d_cin1_presplit(x, g1, b1, 1e-5, wp, 8, g8, b8, 8, 1e-5)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print('cin1 presplit x %d: %.1f us  (%.2f TB/s of input + output)' % (n, us, (n * 4096 * 4 * 9) / us / 1e6))
