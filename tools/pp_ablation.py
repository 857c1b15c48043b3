"""Dev tool: where does k_conv3_up_split_pp (csrc/conv3d_up_split.hip) spend its time?  One-flag variants of the library (-DRF_PP_ABL=<bits>: 1 no staging,
2 no epilogue, 4 no weight loads, 8 no MFMAs; wrong results), built in the CPU container (`python tools/pp_ablation.py build`: only conv3d_up_split.hip is
recompiled per variant) and timed on the GPU box (`python tools/pp_ablation.py [n]`) on the dominant launch, interleaved."""
import os, shutil, subprocess, sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
CSRC = REPO / 'retrieval-fuse_amd' / 'csrc'
OUT = REPO / 'tools' / '_haz'
VARIANTS = {'base': [], 'nostage': ['-DRF_PP_ABL=1'], 'noepi': ['-DRF_PP_ABL=2'], 'noweights': ['-DRF_PP_ABL=4'], 'nomfma': ['-DRF_PP_ABL=8'],
            'nostage_noepi': ['-DRF_PP_ABL=3'], 'mfma_only': ['-DRF_PP_ABL=7'], 'old': ['-DRF_UP_PP=0']}
if len(sys.argv) > 1 and sys.argv[1] == 'build':
    sys.path.insert(0, str(CSRC))
    import build
    build.build()
    for tag, flags in VARIANTS.items():
        od = OUT / ('obj_abl_' + tag)
        od.mkdir(parents=True, exist_ok=True)
        for o in (CSRC / 'build').glob('*.o'):
            if o.name != 'conv3d_up_split.o':
                shutil.copy2(o, od / o.name)
        (od / 'conv3d_up_split.o').unlink(missing_ok=True)
        print(build.build(out=OUT / ('libabl_%s.so' % tag), extra_flags=flags, objdir=od))
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else '8192'
for rnd in range(2):
    for tag in VARIANTS:
        env = dict(os.environ, RFUSE_LIB=str(OUT / ('libabl_%s.so' % tag)))
        r = subprocess.run([sys.executable, str(REPO / 'tools' / 'pp_bench.py'), n] + ([] if tag == 'old' else ['--pm']), env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('n=')]
        print('%-14s %s' % (tag, line[0] if line else r.stderr[-300:]))
