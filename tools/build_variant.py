"""Dev tool (CPU container): a variant of librfuse_hip.so with extra -D flags, for same-box A/B runs through RFUSE_LIB (tools/abn_bench.sh): only the named sources
are recompiled, the other objects are copied from the product build.   python tools/build_variant.py <tag> <source.hip[,source.hip]> [-DFLAG ...]  -> tools/_haz/lib<tag>.so"""
import shutil, sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
CSRC = REPO / 'retrieval-fuse_amd' / 'csrc'
sys.path.insert(0, str(CSRC))
import build
tag, sources, flags = sys.argv[1], sys.argv[2].split(','), sys.argv[3:]
build.build()
od = REPO / 'tools' / '_haz' / ('obj_' + tag)
od.mkdir(parents=True, exist_ok=True)
for o in (CSRC / 'build').glob('*.o'):
    if o.name.replace('.o', '.hip') not in sources:
        shutil.copy2(o, od / o.name)
for src in sources:
    (od / src.replace('.hip', '.o')).unlink(missing_ok=True)
print(build.build(out=REPO / 'tools' / '_haz' / ('lib%s.so' % tag), extra_flags=flags, objdir=od))
