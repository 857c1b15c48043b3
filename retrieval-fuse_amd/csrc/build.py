"""Builds librfuse_hip.so for gfx950 with hipcc (cross-compiles without a GPU).  In-tree output:
retrieval-fuse_amd/rfuse/librfuse_hip.so (git-ignored, travels to the GPU box with the snapshot)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
OUT = HERE.parent / 'rfuse' / 'librfuse_hip.so'
SOURCES = ['capi.hip', 'conv3d.hip', 'conv3d_mfma.hip', 'conv3d_up.hip', 'conv3d_up_split.hip', 'conv3d_split.hip', 'conv3d_small.hip', 'conv3d_backward.hip', 'conv_valid_mfma.hip', 'conv_valid_split.hip', 'linear.hip', 'attention.hip', 'attention_fused.hip', 'retrieval.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
# the attention / gather-normalise kernels restate torch expressions op by op: no a*b+c fusion across operations (explicit fmaf() stays an FMA).
# hipcc's default -ffp-contract=fast fuses in the backend, where neither __fmul_rn nor `#pragma clang fp contract(off)` reach.
EXTRA_FLAGS = {'attention.hip': ['-ffp-contract=off'], 'attention_fused.hip': ['-ffp-contract=off'], 'retrieval.hip': ['-ffp-contract=off']}


def _stale(obj, deps):
    return (not obj.exists()) or any(d.stat().st_mtime > obj.stat().st_mtime for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objdir = HERE / 'build'
    objdir.mkdir(exist_ok=True)
    headers = [HERE / 'common.h', HERE / 'conv_box.h', HERE / 'attn_row.h', HERE.parents[1] / 'include' / 'rfuse.h']

    def compile_one(src):
        obj = objdir / (src.replace('.hip', '.o'))
        if force or _stale(obj, [HERE / src, HERE / 'build.py'] + headers):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ['-c', str(HERE / src), '-o', str(obj)]
            if verbose:
                print(' '.join(cmd))
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(OUT, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(OUT)] + [str(o) for o in objs]
        if verbose:
            print(' '.join(cmd))
        subprocess.run(cmd, check=True)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
