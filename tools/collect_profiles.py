"""Copy the round-end measurements of tools/final_profiles.sh (gpurun_out/final/, merged back by gpurun) into profiles/<tag>_* -- the tracked, committed copies
DESIGN.md / README.md / bench.py cite.    python tools/collect_profiles.py [tag]"""
import shutil
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
src, dst = REPO / 'gpurun_out' / 'final', REPO / 'profiles'
for name in ('bench_C2_default', 'bench_C2_force_collectives', 'bench_C2_isotropic_db', 'bench_C2_fp32_store', 'bench_C1_B16', 'bench_C3_1M', 'bench_C4_B16', 'bench_C5_B16', 'train_C3_B4'):
    f = src / (name + '.json')
    if f.exists() and f.stat().st_size:
        shutil.copy(f, dst / ('%s_%s.json' % (tag, name)))
        print('copied', f.name)
for cfg in ('C2', 'C3', 'C5'):
    hits = list((src / ('prof_' + cfg)).rglob('*kernel_stats.csv'))
    if hits:
        shutil.copy(hits[0], dst / ('%s_bench_%s_kernel_stats.csv' % (tag, cfg)))
        print('copied', hits[0].name)
if (src / 'dbbuild.log').exists():
    shutil.copy(src / 'dbbuild.log', dst / ('%s_dbbuild.log' % tag))
