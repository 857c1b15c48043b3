"""Dev tool (GPU box): the pipelined step of a config with the exact top-k on each scan (auto / VALU / f16-MFMA-filtered), alternating in one process."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd')]
import numpy as np, torch
import bench
from rfuse import configs, synthetic, ops
from rfuse.database import PatchDatabase
from rfuse.engine import RefinementEngine
name = sys.argv[1] if len(sys.argv) > 1 else 'C2'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = configs.get_config(name)
device = torch.device('cuda:0')
torch.manual_seed(0)
emb, meta, vols = bench.synthetic_database(cfg, cfg['db_patches'], device)
db = PatchDatabase(emb, meta, vols, device, 0, 1)
eng = RefinementEngine(cfg, device, db)
raws = torch.from_numpy(np.stack([synthetic.make_chunk(10_000 + b, cfg)['input_raw'] for b in range(B)])).to(device)
def pipe():
    for _ in eng.refine_stream(raws for _ in range(20)): pass
def timed():
    pipe(); torch.cuda.synchronize(); t0 = time.perf_counter(); pipe(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / 20 * 1e3
real_keys, real = ops.l2_topk_keys, ops.l2_topk
def force(algo):
    ops.l2_topk_keys = lambda q, dbp, n, rb, k2, algo_=None, **kw: real_keys(q, dbp, n, rb, k2, algo=algo)
    ops.l2_topk = lambda q, dbp, n, rb, k2, algo_=None, **kw: real(q, dbp, n, rb, k2, algo=algo)
for rep in range(3):
    res = []
    for algo in (ops.TOPK_VALU_SCAN, ops.TOPK_MFMA16_SCAN):
        force(algo); res.append(timed())
    print('%s B=%d  VALU scan %.3f ms   f16-MFMA-filtered scan %.3f ms' % (name, B, res[0], res[1]), flush=True)
