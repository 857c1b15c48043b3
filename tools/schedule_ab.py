"""A config's step under three schedules: pipelined refine_stream, refine() with the side stream, refine() serial."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd')]
import numpy as np, torch
import bench
from rfuse import configs, synthetic
from rfuse.database import PatchDatabase
from rfuse.engine import RefinementEngine
name = sys.argv[1] if len(sys.argv) > 1 else 'C5'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = 10
cfg = configs.get_config(name)
device = torch.device('cuda:0')
torch.manual_seed(0)
emb, meta, vols = bench.synthetic_database(cfg, cfg['db_patches'], device)
db = PatchDatabase(emb, meta, vols, device, 0, 1)
eng = RefinementEngine(cfg, device, db)
raws = torch.from_numpy(np.stack([synthetic.make_chunk(10_000 + b, cfg)['input_raw'] for b in range(B)])).to(device)
def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
def pipe():
    for _ in eng.refine_stream(raws for _ in range(4)): pass
for rep in range(2):
    eng.serial = False
    a = timed(pipe) / 4
    b = timed(lambda: eng.refine(raws))
    eng.serial = True
    c = timed(lambda: eng.refine(raws))
    print('%s B=%d  pipelined %.3f ms  two-stream refine %.3f ms  serial refine %.3f ms' % (name, B, a, b, c), flush=True)
