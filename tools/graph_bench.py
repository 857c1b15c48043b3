"""Dev tool (GPU box): one refine() step of a config at batch B, launched kernel by kernel vs replayed from a captured HIP graph.
    python tools/graph_bench.py [C2] [B=32]"""
import sys, time
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd')]
import bench
from rfuse import configs, synthetic
from rfuse.database import PatchDatabase
from rfuse.engine import RefinementEngine

name = sys.argv[1] if len(sys.argv) > 1 else 'C2'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = configs.get_config(name)
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
torch.manual_seed(0)
emb, meta, vols = bench.synthetic_database(cfg, 50_000, dev)
eng = RefinementEngine(cfg, dev, PatchDatabase(emb, meta, vols, dev, 0, 1))
raws = torch.from_numpy(np.stack([synthetic.make_chunk(10_000 + b, cfg)['input_raw'] for b in range(B)])).to(dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ms = timed(lambda: eng.refine(raws))
print(f'{name} B={B}: launched {ms:.3f} ms/step = {B / ms * 1e3:.0f} chunks/s', flush=True)
eng.serial = True
mss = timed(lambda: eng.refine(raws))
eng.serial = False
print(f'{name} B={B}: one stream (engine.serial) {mss:.3f} ms/step = {B / mss * 1e3:.0f} chunks/s', flush=True)
graph, sin, sout = eng.capture_graph(raws)
ref = eng.refine(raws) if not cfg['attn_retrieval_mode'] else None
msg = timed(graph.replay)
print(f'{name} B={B}: graph replay {msg:.3f} ms/step = {B / msg * 1e3:.0f} chunks/s', flush=True)
if ref is not None:
    graph.replay(); torch.cuda.synchronize()
    print('replayed output equals launched output:', torch.equal(sout, ref))
