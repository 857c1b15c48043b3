"""Dev tool (GPU box): a few launches of the exact top-k over one database size, for profiler passes (tools/pmc_kernel.sh k_l2_topk_mfma16 python tools/topk_one.py 1000000).
    python tools/topk_one.py [n] [nq] [algo] [clustered]"""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
algo = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(nq, 64, generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
emb = torch.randn(n, 64, generator=g, device=dev); emb /= emb.norm(dim=1, keepdim=True)
packed = ops.db_pack_embeddings(emb)
for _ in range(4):
    ops.l2_topk(q, packed, n, 0, 8, algo)
torch.cuda.synchronize()
