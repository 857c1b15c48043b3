"""Dev tool (GPU box): query-side embedding time of every config with the conv patch encoder on the windows / on the padded chunk."""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd'), str(REPO)]
from rfuse import ops
import numpy as np
import bench
from rfuse import configs, synthetic
from rfuse.database import PatchDatabase
from rfuse.engine import RefinementEngine

dev = torch.device('cuda:0')


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, B in (('C1', 16), ('C2', 32), ('C3', 32), ('C4', 16), ('C5', 16)):
    cfg = configs.get_config(name)
    torch.manual_seed(0)
    emb, meta, vols = bench.synthetic_database(cfg, 4096, dev)
    eng = RefinementEngine(cfg, dev, PatchDatabase(emb, meta, vols, dev, 0, 1))
    raw = torch.from_numpy(np.stack([synthetic.make_chunk(10_000 + b, cfg)['input_raw'] for b in range(B)])).to(dev)
    g = eng.config['query_geometry']
    enc = eng.fenc_input
    plan = enc.grid_plan(g['patch_size_input'] + 2 * g['patch_context_input'], g['patch_size_input'], raw.shape[-1] // g['patch_size_input']) if hasattr(enc, 'grid_plan') else None
    res = {}
    for flag in (False, True):
        ops.USE_FCN_ENCODER = flag
        res[flag] = timed(lambda: eng.embed_queries(raw))
    ops.USE_FCN_ENCODER = True
    a, b = eng.embed_queries(raw), None
    ops.USE_FCN_ENCODER = False
    b = eng.embed_queries(raw)
    ops.USE_FCN_ENCODER = True
    print('%s (B = %d, %s, windows %d+2x%d of a %d^3 chunk, plan %s): windows %.3f ms, grid %.3f ms, max |diff| %.1e' % (
        name, B, type(enc).__name__, g['patch_size_input'], g['patch_context_input'], raw.shape[-1], plan, res[False], res[True], float((a - b).abs().max())), flush=True)
