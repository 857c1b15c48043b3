"""Dev tool (GPU box): the pipelined C2 step with the back end on a HIGH-priority stream (front end and chunk-level U-Net on normal-priority helper streams) against
the default (everything normal priority, back end on the null stream), alternating in one process."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd')]
import numpy as np, torch
import bench
from rfuse import configs, synthetic
from rfuse.database import PatchDatabase
from rfuse.engine import RefinementEngine
name = sys.argv[1] if len(sys.argv) > 1 else 'C2'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = configs.get_config(name)
device = torch.device('cuda:0')
torch.manual_seed(0)
print('priority range', torch.cuda.Stream.priority_range())
emb, meta, vols = bench.synthetic_database(cfg, cfg['db_patches'], device)
eng = RefinementEngine(cfg, device, PatchDatabase(emb, meta, vols, device, 0, 1))
batches = [torch.from_numpy(np.stack([synthetic.make_chunk(10_000 + r * B + b, cfg)['input_raw'] for b in range(B)])).to(device) for r in range(4)]
hi = torch.cuda.Stream(device, priority=-1)
plain = torch.cuda.Stream(device)
def pipe(n=40):
    for _ in eng.refine_stream(batches[i % 4] for i in range(n)): pass
def timed(stream):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.default_stream())
    with ctx:
        pipe(10); torch.cuda.synchronize(); t0 = time.perf_counter(); pipe(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / 40 * 1e3
for rep in range(4):
    print('%s B=%d  null stream %.3f ms   pool stream (normal priority) %.3f ms   high-priority stream %.3f ms' % (name, B, timed(None), timed(plain), timed(hi)), flush=True)
