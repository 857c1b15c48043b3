"""Dev tool: host-side enqueue time of one engine step (is the GPU ever waiting for Python?)."""
import os, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd')]
import numpy as np, torch
import bench
from rfuse import configs, synthetic
from rfuse.database import PatchDatabase
from rfuse.engine import RefinementEngine
dev = torch.device('cuda', 0)
cfg = configs.get_config('C2'); B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
emb, meta, vols = bench.synthetic_database(cfg, 50000, dev)
db = PatchDatabase(emb, meta, vols, dev, 0, 1)
eng = RefinementEngine(cfg, dev, db)
raws = np.stack([synthetic.make_chunk(10_000 + b, cfg)['input_raw'] for b in range(B)])
raw = torch.from_numpy(raws).to(dev)
for _ in range(3): eng.refine(raw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): eng.refine(raw)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue %.2f ms/step   total %.2f ms/step' % ((t1 - t0) * 100, (t2 - t0) * 100))
