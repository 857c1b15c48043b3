"""Dev tool (GPU box): every C-ABI launch of one serial step with its HIP-event time (the bench's kernel table, untruncated).
    python tools/step_profile.py [config] [batch]"""
import sys
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
dev = torch.device('cuda:0')
torch.manual_seed(0)
emb, meta, vols = bench.synthetic_database(cfg, 50_000, dev)
eng = RefinementEngine(cfg, dev, PatchDatabase(emb, meta, vols, dev))
raw = torch.from_numpy(np.stack([synthetic.make_chunk(10_000 + b, cfg)['input_raw'] for b in range(B)])).to(dev)
t, _ = bench.kernel_table(eng, raw, cfg, steps=3, top=60)
print('serial ms/step %.3f' % t['serial_ms_per_step'])
for r in t['top']:
    print('%-30s %-34s x%-3.0f %7.3f ms  %5.1f%%  %s' % (r['entry'], r['args'], r['launches_per_step'], r['ms_per_step'], 100 * r['share'],
          ('%s %.1f/%.0f %s = %.2f' % (r['bound'], r['achieved'], r['peak'], r['unit'], r['frac'])) if 'bound' in r else ''))
