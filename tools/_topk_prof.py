import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(2048, 64, generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
n = 50001
emb = torch.randn(n, 64, generator=g, device=dev); emb /= emb.norm(dim=1, keepdim=True)
packed = ops.db_pack_embeddings(emb)
for algo in (1, 3):
    for _ in range(20):
        ops.l2_topk(q, packed, n, 0, 8, algo)
torch.cuda.synchronize()
