"""Dev tool (GPU box): time the exact top-k scans (VALU scan / fp32-MFMA-filtered / split-f16-MFMA-filtered) over database sizes, on isotropic unit Gaussians and on
CLUSTERED embeddings (rows = a random query + noise of the queries' nearest-neighbour spacing: a database that lies where the queries lie -- bench.py's construction).
    python tools/topk_bench.py [nq] [k2]"""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops
dev = torch.device('cuda:0')
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
K2 = int(sys.argv[2]) if len(sys.argv) > 2 else 8
g = torch.Generator(device=dev).manual_seed(0)
q_iso = torch.randn(nq, 64, generator=g, device=dev); q_iso /= q_iso.norm(dim=1, keepdim=True)
# clustered queries: all within ~0.3 of one direction (a random-init encoder's embeddings; squared nearest-neighbour distance ~8e-3)
q_clu = torch.randn(1, 64, generator=g, device=dev) + 0.04 * torch.randn(nq, 64, generator=g, device=dev); q_clu /= q_clu.norm(dim=1, keepdim=True)
for kind, n in [(k, n) for k in ('isotropic', 'clustered') for n in (12_500, 50_000, 125_000, 250_000, 1_000_000)]:
    if kind == 'isotropic':
        q = q_iso
        emb = torch.randn(n, 64, generator=g, device=dev)
    else:
        q = q_clu
        d2 = (2 - 2 * q[:512] @ q.T).clamp_min(0); d2[d2 < 1e-9] = 9
        sigma = d2.min(dim=1).values.median().sqrt().item() / 8
        emb = q[torch.randint(0, nq, (n,), generator=g, device=dev)] + sigma * torch.randn(n, 64, generator=g, device=dev)
    emb /= emb.norm(dim=1, keepdim=True)
    packed = ops.db_pack_embeddings(emb)
    out = {}
    for algo in (1, 2, 3):
        for _ in range(2):
            r = ops.l2_topk(q, packed, n, 0, K2, algo)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            r = ops.l2_topk(q, packed, n, 0, K2, algo)
        e1.record(); torch.cuda.synchronize()
        out[algo] = (e0.elapsed_time(e1) / 5, r)
    same = all(torch.equal(out[1][1][1], out[a][1][1]) and torch.equal(out[1][1][0], out[a][1][0]) for a in (2, 3))
    pairs = nq * n
    print('%-9s n=%8d nq=%d  VALU scan %.3f ms (%.1f Gpair/s)   fp32-MFMA filter %.3f ms (%.1f TF/s of q.x)   f16-MFMA filter %.3f ms (%.1f Gpair/s)  identical=%s'
          % (kind, n, nq, out[1][0], pairs / out[1][0] / 1e6, out[2][0], pairs * 128 / out[2][0] / 1e9, out[3][0], pairs / out[3][0] / 1e6, same))
