"""Dev tool (GPU box): how many of a step's K * 64 * B retrieved database rows are distinct?  (The retrieval backbone's output depends on the patch alone.)"""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd')]
import numpy as np, torch
import bench
from rfuse import configs, synthetic
from rfuse.database import PatchDatabase
from rfuse.engine import RefinementEngine
for name, B in (('C2', 32), ('C3', 32)):
    cfg = configs.get_config(name)
    device = torch.device('cuda:0')
    torch.manual_seed(0)
    emb, meta, vols = bench.synthetic_database(cfg, 50000, device)
    eng = RefinementEngine(cfg, device, PatchDatabase(emb, meta, vols, device, 0, 1))
    for r in range(3):
        raw = torch.from_numpy(np.stack([synthetic.make_chunk(10_000 + r * B + b, cfg)['input_raw'] for b in range(B)])).to(device)
        q = eng.embed_queries(raw)
        _, _, idx = eng.database.retrieve(q, eng.K)
        u = torch.unique(idx).numel()
        print(name, 'batch', r, 'retrieved', idx.numel(), 'distinct rows', u, '(%.1f %%)' % (100.0 * u / idx.numel()))
