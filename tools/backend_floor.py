"""Dev tool (GPU box): C2 step pipelined (refine_stream) vs the back end alone (retrieval backbone + attention + decoder on fixed front-end results) vs the
front end alone -- what sharing the GPU between the two costs."""
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
N = 40
def pipe(n=N):
    for _ in eng.refine_stream(batches[i % 4] for i in range(n)): pass
def t(fn):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / N * 1e3
with torch.no_grad():
    x_in = eng.normalise_input(batches[0]); patches, _ = eng.retrieve(batches[0]); x_back = eng.unet_backbone(x_in)
    def back():
        for _ in range(N):
            eng._attend_and_decode(x_back, eng.retrieval_backbone(patches), None)
    def front():
        for i in range(N):
            xb, side = eng._fork_backbone(eng.normalise_input(batches[i % 4])); eng.retrieve(batches[i % 4]); torch.cuda.current_stream().wait_stream(side)
    def front_noback():
        for i in range(N):
            eng.retrieve(batches[i % 4])
    def plain():
        for i in range(N): eng.refine(batches[i % 4])
    for rep in range(3):
        print('pipelined %.3f  refine() %.3f  back end alone %.3f  front end alone (with backbone) %.3f  retrieve alone %.3f ms' % (t(pipe), t(plain), t(back), t(front), t(front_noback)), flush=True)
