"""Dev tool (CPU container): builds VARIANTS of the fp32 box conv (csrc/conv3d_mfma.hip) as separate shared objects under tools/_haz/
for tools/hazard_probe2.py (GPU box).  Each variant is the product source with one small textual patch -- the product source itself carries
no switches.  Question behind it: why do the fp32 LDS-DMA conv kernels return different last bits when a
kernel that issues F16 MFMAs shares their SIMD?

    python tools/hazard_variants.py            # writes tools/_haz/libv_<name>.so + tools/_haz/aggr.so
"""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
CSRC = REPO / 'retrieval-fuse_amd' / 'csrc'
OUT = REPO / 'tools' / '_haz'
HIPCC = '/opt/rocm/bin/hipcc'
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-function', '-I', str(CSRC)]

SYNC_A = '        __syncthreads();                                   // everyone is done reading xs / ws[buf]; loads + DMA have landed\n'
DMA = '                    __builtin_amdgcn_global_load_lds((rf_gptr)(src + off), (rf_lptr)(dst + q * 256), 16, 0, 0);\n'
PFX = 'auto kern = k_conv3_mfma<TZ, TY, TX, SPW, NW, MB, NB, WPS, true, CCT>;'
COMMIT_STORE = '''                float* dst = xs + rowbox[i];
#pragma unroll
                for (int j = 0; j < TX + 2; ++j) dst[j] = v[j];
'''
DUMP_DECL = '''__device__ float* g_dbg = nullptr;
extern "C" int rf_dbg_set(float* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &p, sizeof(p)); }
'''

# name -> (list of (old, new) patches, extra flags)
VARIANTS = {
    'control': ([], []),
    # nothing may move across the end of the MFMA loop (hipcc sinks the last step's MFMAs below the barrier, next to the commit's VALU code)
    'schedbar': ([(SYNC_A, '        __builtin_amdgcn_sched_barrier(0);\n' + SYNC_A)], []),
    # ... and 64 idle cycles between the last MFMA and the barrier / the commit's VALU instructions
    'nops': ([(SYNC_A, '        __builtin_amdgcn_sched_barrier(0);\n        asm volatile("s_nop 15\\n s_nop 15\\n s_nop 15\\n s_nop 15" ::: "memory");\n'
               '        __builtin_amdgcn_sched_barrier(0);\n' + SYNC_A)], []),
    # weight slab through registers instead of LDS-DMA
    'nodma': ([(DMA, '                    *reinterpret_cast<float4*>(dst + q * 256 + lane * 4) = *reinterpret_cast<const float4*>(src + off);\n')], []),
    # input rows loaded after the MFMA loop (no prefetch registers live across it)
    'nopfx': ([(PFX, PFX.replace('true', 'false'))], []),
    # no packed fp32 VALU forms in the commit (v_pk_add_f32 / v_pk_fma_f32 come from the SLP vectoriser)
    'noslp': ([], ['-fno-slp-vectorize']),
    # control + the committed (GroupNorm-applied) halo rows are also written to a global debug buffer
    'dump': ([(COMMIT_STORE, COMMIT_STORE + '''                if (g_dbg) {
                    float* d = g_dbg + ((((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 16 + (size_t)(cbase >> 2)) * (size_t)(RPT * NT) + (size_t)(tid + i * NT)) * 16;
#pragma unroll
                    for (int j = 0; j < TX + 2; ++j) d[j] = v[j];
                }
'''), ('typedef __attribute__((address_space(1))) const void* rf_gptr;', DUMP_DECL + 'typedef __attribute__((address_space(1))) const void* rf_gptr;')], []),
}


def build_variant(name):
    patches, extra = VARIANTS[name]
    src = (CSRC / 'conv3d_mfma.hip').read_text()
    for old, new in patches:
        assert src.count(old) == 1, (name, old)
        src = src.replace(old, new)
    p = OUT / ('conv3d_mfma_%s.hip' % name)
    p.write_text(src)
    so = OUT / ('libv_%s.so' % name)
    objs = [str(CSRC / 'build' / o) for o in ('capi.o', 'conv3d_small.o')]
    obj = OUT / ('conv3d_mfma_%s.o' % name)
    subprocess.run([HIPCC] + FLAGS + extra + ['-c', str(p), '-o', str(obj)], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(so), str(obj)] + objs, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return so


def main():
    OUT.mkdir(exist_ok=True)
    sys.path.insert(0, str(CSRC))
    import build as product_build
    product_build.build()
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', str(OUT / 'aggr.so'), str(REPO / 'tools' / 'micro' / 'hazard_probe.hip')], check=True)
    with ThreadPoolExecutor(max_workers=8) as ex:
        for so in ex.map(build_variant, list(VARIANTS)):
            print(so)


if __name__ == '__main__':
    main()
