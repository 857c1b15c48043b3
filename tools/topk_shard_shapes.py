"""Dev tool (GPU box): the exact top-16 of the C2 bench as ONE rank of an N-way sharded database sees it (weak scaling: N * 2048 queries against 50 k / N rows) --
the per-rank scan cost that 1 / 2 / 4 / 8-GPU runs add to a step -- for the scan the library picks (algo 0) and the two forced ones."""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
total = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
for N in (1, 2, 4, 8):
    nq, n = 2048 * N, total // N
    q = torch.randn(nq, 64, generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
    emb = torch.randn(n, 64, generator=g, device=dev); emb /= emb.norm(dim=1, keepdim=True)
    packed = ops.db_pack_embeddings(emb)
    res = {}
    for algo in (0, 1, 3):
        for _ in range(2): ops.l2_topk_keys(q, packed, n, 0, 8, algo)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.l2_topk_keys(q, packed, n, 0, 8, algo)
        e1.record(); torch.cuda.synchronize()
        res[algo] = e0.elapsed_time(e1) / 5
    print('N=%d: %5d queries x %6d rows   picked %.3f ms   VALU scan %.3f   f16-MFMA filter %.3f' % (N, nq, n, res[0], res[1], res[3]))
