"""Dev tool (GPU box): the pipelined C2 step with the next batch's front end issued at the start of the back end vs behind its encoders, alternating in one process."""
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
emb, meta, vols = bench.synthetic_database(cfg, cfg['db_patches'], device)
eng = RefinementEngine(cfg, device, PatchDatabase(emb, meta, vols, device, 0, 1))
batches = [torch.from_numpy(np.stack([synthetic.make_chunk(10_000 + r * B + b, cfg)['input_raw'] for b in range(B)])).to(device) for r in range(4)]
def pipe(n=40):
    for _ in eng.refine_stream(batches[i % 4] for i in range(n)): pass
def timed():
    pipe(10); torch.cuda.synchronize(); t0 = time.perf_counter(); pipe(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / 40 * 1e3
for rep in range(4):
    res = []
    for mode in ('start', 'unet_decoders'):
        eng.front_at = mode; res.append(timed())
    print('%s B=%d  front at start %.3f ms   U-Net behind its own encoders %.3f ms' % (name, B, res[0], res[1]), flush=True)
eng.front_at = 'start'; a = [x.clone() for x in eng.refine_stream(batches)]
eng.front_at = 'unet_decoders'; b = [x.clone() for x in eng.refine_stream(batches)]
print('bit-equal:', all(torch.equal(x, y) for x, y in zip(a, b)))
