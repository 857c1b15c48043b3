"""Dev tool (GPU box): the fused attention encoder (four Linear layers, split-operand form, k_attn_mlp_split) alone on the C2 step's two launches:
the retrieved volumes (32 chunks x K = 4 x 16 channels @32^3 in 8^3 patches -> 524,288 rows) and the backbone volumes (32 x 1 -> 131,072 rows).

    python tools/attn_mlp_bench.py"""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops

dev = torch.device('cuda:0')
torch.manual_seed(2)
params = []
for o, i in ((128, 128), (128, 128), (128, 128), (32, 128)):
    params += [0.1 * torch.randn(o, i, device=dev), 0.1 * torch.randn(o, device=dev)]
packed = ops.pack_attn_mlp(params)
for b, kv, c, s, t in ((32, 4, 16, 32, 8), (32, 1, 16, 32, 32)):
    src = torch.randn(b * kv * (s // t) ** 3, c, t, t, t, device=dev)
    run = lambda: ops.attn_mlp_volume(src, b, kv, c, s, t, packed)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    rows = b * kv * (s // 2) ** 3
    flop = rows * 2.0 * 3 * (128 * 128 * 3 + 128 * 32)
    print('[%d, %d, %d, %d, %d]  %d rows  %.3f ms  %.0f TFLOP/s issued (%.2f of 2500)' % (b, kv, c, s, t, rows, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500))
    if '--digest' in sys.argv:
        import hashlib
        print('   digest', hashlib.sha256(run().cpu().numpy().tobytes()).hexdigest()[:16])
