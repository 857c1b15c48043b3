"""Dev tool (GPU box): time the two exact top-k scans (VALU scan / MFMA-filtered scan) over database sizes.
    python tools/topk_bench.py [nq]"""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops
dev = torch.device('cuda:0')
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(nq, 64, generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
for n in (12_500, 50_000, 125_000, 250_000, 1_000_000):
    emb = torch.randn(n, 64, generator=g, device=dev); emb /= emb.norm(dim=1, keepdim=True)
    packed = ops.db_pack_embeddings(emb)
    out = {}
    for algo in (1, 2, 3):
        for _ in range(2):
            r = ops.l2_topk(q, packed, n, 0, 8, algo)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            r = ops.l2_topk(q, packed, n, 0, 8, algo)
        e1.record(); torch.cuda.synchronize()
        out[algo] = (e0.elapsed_time(e1) / 5, r)
    same = all(torch.equal(out[1][1][1], out[a][1][1]) and torch.equal(out[1][1][0], out[a][1][0]) for a in (2, 3))
    pairs = nq * n
    print('n=%8d nq=%d  VALU scan %.3f ms (%.1f Gpair/s)   fp32-MFMA filter %.3f ms (%.1f TF/s of q.x)   f16-MFMA filter %.3f ms (%.1f Gpair/s)  identical=%s'
          % (n, nq, out[1][0], pairs / out[1][0] / 1e6, out[2][0], pairs * 128 / out[2][0] / 1e9, out[3][0], pairs / out[3][0] / 1e6, same))
