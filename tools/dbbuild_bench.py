"""Dev tool (GPU box): throughput of the database build path: windows -> fenc_target (Patch32 for the super-resolution
configs, Patch24V2 for surface reconstruction) -> L2 normalise, in 64^3 chunks (= 64 database patches) per second.
    python tools/dbbuild_bench.py [C1|C5] [chunks]"""
import contextlib, io, sys, time
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd')]
import model
from rfuse import configs, synthetic
from rfuse.database import build_database_rows
cfg = configs.get_config(sys.argv[1] if len(sys.argv) > 1 else 'C1')
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    _, fenc_target = model.get_retrieval_networks(cfg['retrieval_model'])
vols = torch.from_numpy(np.stack([synthetic.make_chunk(i, cfg)['target_raw'] for i in range(32)])).repeat(n // 32, 1, 1, 1).to(dev)
for cpb in (8, 32):
    build_database_rows(cfg, fenc_target, vols[:cpb], dev, chunks_per_batch=cpb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    emb, meta = build_database_rows(cfg, fenc_target, vols, dev, chunks_per_batch=cpb)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%s %s: %d chunks (%d patches) in %.3f s = %.0f chunks/s = %.0f patches/s  (batches of %d chunks)'
          % (sys.argv[1] if len(sys.argv) > 1 else 'C1', type(fenc_target).__name__, n, emb.shape[0] - 1, dt, n / dt, (emb.shape[0] - 1) / dt, cpb))
