"""Dev tool: where does the fused attention encoder (k_attn_mlp_split, csrc/attention_fused.hip) spend its time?  One-patch variants of the whole
library, built in the CPU container (`python tools/attn_mlp_ablation.py build`), timed on the GPU box (`python tools/attn_mlp_ablation.py`) through
tools/attn_mlp_bench.py.  Variant results are wrong on purpose; timing only."""
import os, subprocess, sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
CSRC = REPO / 'retrieval-fuse_amd' / 'csrc'
OUT = REPO / 'tools' / '_haz'
GATHER = """                    const float2 lo2 = *reinterpret_cast<const float2*>(p + (size_t)(2 * kb) * t3);
                    const float2 hi2 = *reinterpret_cast<const float2*>(p + (size_t)(2 * kb) * t3 + t);"""
NO_GATHER = """                    const float2 lo2 = make_float2((float)(kb + g), (float)j);
                    const float2 hi2 = make_float2((float)row, (float)(kb - g));"""
EPI = """                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * 0.01f;
                    ams_split4(v, bh[ib >> 1], bl[ib >> 1], 4 * (ib & 1));"""
NO_EPI = """                    for (int e = 0; e < 1; ++e) v[e] += 1.f;
                    if (v[0] == 123.456f) ams_split4(v, bh[ib >> 1], bl[ib >> 1], 4 * (ib & 1));"""
MF = """                hi[ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[cur], bh[t], hi[ib], 0, 0, 0);
                lo[ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[cur], bl[t], lo[ib], 0, 0, 0);
                lo[ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[cur], bh[t], lo[ib], 0, 0, 0);"""
NO_MF = """                hi[ib][0] += (float)wh[cur][0] * (float)bh[t][0]; lo[ib][0] += (float)wl[cur][0] * (float)bl[t][0];"""
STORE = "                        *reinterpret_cast<float4*>(a.out + orow * AM_OUT + ib * 16 + 4 * g) = o;"
NO_STORE = "                        if (o.x == 123.456f) *reinterpret_cast<float4*>(a.out + orow * AM_OUT + ib * 16 + 4 * g) = o;"
DMA = "            if (layer < 3) dma_layer(layer + 1);\n            else if (wt + (int)gridDim.x < nwt) dma_layer(0);"
NO_DMA = "            if (layer < 3 && wt < 0) dma_layer(layer + 1);"
BAR = "            __syncthreads();                                          // weights of `layer` landed; everyone left layer-1's buffer"
NO_BAR = ""
VARIANTS = {'base': [], 'no_gather': [(GATHER, NO_GATHER)], 'no_epilogue': [(EPI, NO_EPI)], 'no_mfma': [(MF, NO_MF)], 'no_store': [(STORE, NO_STORE)],
            'no_dma': [(DMA, NO_DMA)], 'no_dma_no_barrier': [(DMA, NO_DMA), (BAR, NO_BAR)],
            'mfma_only': [(GATHER, NO_GATHER), (EPI, NO_EPI), (STORE, NO_STORE), (DMA, NO_DMA), (BAR, NO_BAR)]}


def build():
    OUT.mkdir(exist_ok=True)
    objs = [str(p) for p in sorted((CSRC / 'build').glob('*.o')) if p.name != 'attention_fused.o']
    for name, patches in VARIANTS.items():
        src = (CSRC / 'attention_fused.hip').read_text()
        cut = src.index('fused MLP, split-operand form')
        head, src = src[:cut], src[cut:]
        for old, new in patches:
            assert src.count(old) == 1, (name, old[:60], src.count(old))
            src = src.replace(old, new)
        src = head + src
        p = OUT / ('attnmlp_%s.hip' % name)
        p.write_text(src)
        obj = OUT / ('attnmlp_%s.o' % name)
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', str(CSRC), '-c', str(p), '-o', str(obj)], check=True)
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(OUT / ('librfuse_attnmlp_%s.so' % name)), str(obj)] + objs, check=True)
        print(name)


if __name__ == '__main__':
    if sys.argv[1:] == ['build']:
        build()
    else:
        for name in VARIANTS:
            env = dict(os.environ, RFUSE_LIB=str(OUT / ('librfuse_attnmlp_%s.so' % name)))
            r = subprocess.run([sys.executable, str(REPO / 'tools' / 'attn_mlp_bench.py')], env=env, capture_output=True, text=True)
            print('%-18s %s' % (name, [l.split('rows')[1].strip() for l in r.stdout.splitlines() if 'rows' in l] or r.stderr[-300:]))
