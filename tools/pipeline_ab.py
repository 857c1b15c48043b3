"""Dev tool (GPU box): step time of RefinementEngine.refine back to back vs RefinementEngine.refine_stream (front end of batch i+1 beside the back end of batch i)."""
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
cfg = configs.get_config(name); dev = torch.device('cuda:0')
torch.manual_seed(0)
emb, meta, vols = bench.synthetic_database(cfg, cfg['db_patches'] if name != 'C3' else 50_000, dev)
eng = RefinementEngine(cfg, dev, PatchDatabase(emb, meta, vols, dev))
raw = torch.from_numpy(np.stack([synthetic.make_chunk(10_000 + b, cfg)['input_raw'] for b in range(B)])).to(dev)
noise_free = not cfg['attn_retrieval_mode']
for _ in range(3): ref = eng.refine(raw)
outs = list(eng.refine_stream([raw] * 3))
torch.cuda.synchronize()
if noise_free:
    print('bit-identical to refine():', all(torch.equal(o, ref) for o in outs))
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20): eng.refine(raw)
    torch.cuda.synchronize(); a = (time.perf_counter() - t0) * 50
    t0 = time.perf_counter()
    for out in eng.refine_stream(raw for _ in range(20)): pass
    torch.cuda.synchronize(); b = (time.perf_counter() - t0) * 50
    print('%s B=%d: refine %.3f ms/step | refine_stream %.3f ms/step' % (name, B, a, b), flush=True)
