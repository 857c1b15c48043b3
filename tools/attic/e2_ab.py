"""Dev tool (GPU box): the pipelined step of a config with the 2^3 / 1^3 GEMM form on and off, alternating in one process."""
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
real = ops.conv_e2_split_supported
for rep in range(4):
    ops.conv_e2_split_supported = real
    a = timed()
    ops.conv_e2_split_supported = lambda *args: False
    b = timed()
    print('%s B=%d  2^3 GEMM form %.3f ms   fp32 position-major %.3f ms' % (name, B, a, b), flush=True)
