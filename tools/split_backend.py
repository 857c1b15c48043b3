"""Dev tool (GPU box): the C2 back end (retrieval backbone -> attention -> decoder) of a 32-chunk batch as ONE pass against TWO concurrent passes of 16 chunks on
two streams (chunks are independent): does the other half's work fill the tails of the big launches?"""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd')]
import numpy as np, torch
import bench
from rfuse import configs, synthetic
from rfuse.database import PatchDatabase
from rfuse.engine import RefinementEngine
cfg = configs.get_config('C2'); B = 32
device = torch.device('cuda:0')
torch.manual_seed(0)
emb, meta, vols = bench.synthetic_database(cfg, cfg['db_patches'], device)
eng = RefinementEngine(cfg, device, PatchDatabase(emb, meta, vols, device, 0, 1))
raw = torch.from_numpy(np.stack([synthetic.make_chunk(10_000 + b, cfg)['input_raw'] for b in range(B)])).to(device)
N = 40
main = torch.cuda.current_stream(); s1 = torch.cuda.Stream(device)
with torch.no_grad():
    x_in = eng.normalise_input(raw); patches, _ = eng.retrieve(raw); x_back = eng.unet_backbone(x_in)
    per = patches.shape[0] // B
    torch.cuda.synchronize()
    def back(p, xb):
        return eng._attend_and_decode(xb, eng.retrieval_backbone(p), None)
    def one():
        for _ in range(N): back(patches, x_back)
    def split(parts):
        cb = B // parts
        def go():
            for _ in range(N):
                outs = []
                ev = torch.cuda.Event(); ev.record(main)
                for k in range(parts):
                    st = main if k == 0 else s1
                    if k: st.wait_event(ev)
                    with torch.cuda.stream(st):
                        outs.append(back(patches[k * cb * per:(k + 1) * cb * per], x_back[k * cb:(k + 1) * cb]))
                main.wait_stream(s1)
        return go
    def t(fn):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / N * 1e3
    a = back(patches, x_back); torch.cuda.synchronize()
    for rep in range(3):
        print('one pass of 32 chunks %.3f ms | two concurrent passes of 16 %.3f ms' % (t(one), t(split(2))), flush=True)
