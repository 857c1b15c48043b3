"""Dev tool: where does the persistent decoder box kernel (k_conv3_up_split_boxp, csrc/conv3d_up_split.hip) spend its time?  One-patch variants of the
whole library, built in the CPU container (`python tools/upbox_ablation.py build`), timed on the GPU box (`python tools/upbox_ablation.py`) through
tools/upbox_bench.py.  Variant results are wrong on purpose; timing only."""
import os, subprocess, sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
CSRC = REPO / 'retrieval-fuse_amd' / 'csrc'
OUT = REPO / 'tools' / '_haz'
STATS = "            if (co < cout) {\n#pragma unroll 2\n                for (int i = 0; i < 16; ++i) {\n                    const float4 v = *reinterpret_cast<const float4*>(e + co * UB_E_STRIDE + (part * 16 + i) * 4);"
NO_STATS = STATS.replace("if (co < cout) {", "if (co < cout && b < 0) {")
STORES = "        for (int it = 0; it < 2 * CGO; ++it) {\n            const int q = td + it * 512;"
NO_STORES = "        for (int it = 0; it < (b < 0 ? 2 * CGO : 0); ++it) {\n            const int q = td + it * 512;"
LOADS = "        for (int j = 0; j < 8; ++j) xr[j] = *reinterpret_cast<const float*>(sb + (size_t)(off + (unsigned)(j * hvol) * 4u));"
NO_LOADS = "        for (int j = 0; j < 8; ++j) xr[j] = (float)(off + j);"
STAGE = "        if (b + 1 < b1) stage(cur ^ 1);"
NO_STAGE = "        if (b + 1 < b1 && b < 0) stage(cur ^ 1);"
TILE = "                    e[col * UB_E_STRIDE + lin] = fmaxf(fmaf(lo[m][0][r], 1.0f / US_LO, hi[m][0][r]), 0.f);\n                }\n        }\n        if (b + 1 < b1)"
NO_TILE = TILE.replace("e[col * UB_E_STRIDE + lin] =", "if (hi[m][0][r] == 123.456f) e[col * UB_E_STRIDE + lin] =")
MF = "                    us_mfma_block<1>(hi[m], lo[m], ah, al, bh, bl);\n                }\n            }\n        if (b + 1 < b1 && ((b + 1)"
NO_MF = MF.replace("us_mfma_block<1>(hi[m], lo[m], ah, al, bh, bl);", "hi[m][0][0] += (float)ah[0] * (float)bh[0][0]; lo[m][0][0] += (float)al[0] * (float)bl[0][0];")
VARIANTS = {'base': [], 'no_stats': [(STATS, NO_STATS)], 'no_stores': [(STORES, NO_STORES)], 'no_loads': [(LOADS, NO_LOADS)], 'no_stage': [(STAGE, NO_STAGE)],
            'no_tile': [(TILE, NO_TILE)], 'no_mfma': [(MF, NO_MF)],
            'mfma_only': [(STATS, NO_STATS), (STORES, NO_STORES), (LOADS, NO_LOADS), (TILE, NO_TILE), (STAGE, NO_STAGE)]}


def build():
    OUT.mkdir(exist_ok=True)
    objs = [str(p) for p in sorted((CSRC / 'build').glob('*.o')) if p.name != 'conv3d_up_split.o']
    for name, patches in VARIANTS.items():
        src = (CSRC / 'conv3d_up_split.hip').read_text()
        for old, new in patches:
            assert src.count(old) == 1, (name, old[:60], src.count(old))
            src = src.replace(old, new)
        p = OUT / ('upbox_%s.hip' % name)
        p.write_text(src)
        obj = OUT / ('upbox_%s.o' % name)
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', str(CSRC), '-c', str(p), '-o', str(obj)], check=True)
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(OUT / ('librfuse_upbox_%s.so' % name)), str(obj)] + objs, check=True)
        print(name)


if __name__ == '__main__':
    if sys.argv[1:] == ['build']:
        build()
    else:
        for name in VARIANTS:
            env = dict(os.environ, RFUSE_LIB=str(OUT / ('librfuse_upbox_%s.so' % name)))
            r = subprocess.run([sys.executable, str(REPO / 'tools' / 'upbox_bench.py')], env=env, capture_output=True, text=True)
            print('%-10s %s' % (name, [l for l in r.stdout.splitlines() if l.startswith('ch8')] or r.stderr[-300:]))
