"""Dev tool (GPU box): where a refine_scene call spends its time (64 chunks, two batches of 32)."""
import sys, time
from pathlib import Path
import numpy as np
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd')]
import bench
from rfuse import configs, scene, synthetic
from rfuse.database import PatchDatabase
from rfuse.engine import RefinementEngine
cfg_name = sys.argv[1] if len(sys.argv) > 1 else 'C2'
cfg = configs.get_config(cfg_name)
dev = torch.device('cuda:0')
emb, meta, vols = bench.synthetic_database(cfg, 50_000, dev)
eng = RefinementEngine(cfg, dev, PatchDatabase(emb, meta, vols, dev))
trunc_i, _ = configs.truncations(cfg)
grid, s_in = (4, 4, 4), cfg['dataset_train']['input_chunk_size']
base = np.stack([synthetic.make_chunk(30_000 + i, cfg)['input_raw'] for i in range(64)])
low = base.reshape(grid + (s_in,) * 3).transpose(0, 3, 1, 4, 2, 5).reshape(grid[0] * s_in, grid[1] * s_in, grid[2] * s_in)
names, chunks = scene.split_scene(low, s_in, 'bench', pad_value=trunc_i)
x = torch.from_numpy(chunks).pin_memory()
def sync(): torch.cuda.synchronize()
for _ in range(3):
    scene.refine_scene(eng, names, chunks, batch=32)
def timeit(fn, reps=6):
    sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    sync(); return 1e3 * (time.perf_counter() - t0) / reps
print('refine_stream, 2 batches of 32, results dropped      %.2f ms' % timeit(lambda: [None for _ in eng.refine_stream(x[lo:lo + 32].to(dev, non_blocking=True) for lo in (0, 32))]))
print('refine x 2 (one after the other)                     %.2f ms' % timeit(lambda: [eng.refine(x[lo:lo + 32].to(dev, non_blocking=True)) for lo in (0, 32)]))
print('refine_scene (device assembly)                       %.2f ms' % timeit(lambda: scene.refine_scene(eng, names, chunks, batch=32)))
print('refine_scene (host assembly)                         %.2f ms' % timeit(lambda: scene.refine_scene(eng, names, chunks, batch=32, assemble_on_device=False)))
print('refine_scene, batch 64                               %.2f ms' % timeit(lambda: scene.refine_scene(eng, names, chunks, batch=64)))
h = torch.empty(64 * 64 ** 3, dtype=torch.float64, pin_memory=True); d = torch.zeros(64 * 64 ** 3, dtype=torch.float64, device=dev)
print('134 MB device -> pinned host                         %.2f ms' % timeit(lambda: h.copy_(d, non_blocking=True)))
print('pinned allocation of 134 MB (cached)                 %.2f ms' % timeit(lambda: torch.empty(64 * 64 ** 3, dtype=torch.float64, pin_memory=True)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(4): scene.refine_scene(eng, names, chunks, batch=32)
pr.disable(); sync()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
