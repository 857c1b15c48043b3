"""Builds librfuse_hip.so for gfx950 with hipcc (cross-compiles without a GPU).  In-tree output:
retrieval-fuse_amd/rfuse/librfuse_hip.so (git-ignored, travels to the GPU box with the snapshot)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
OUT = HERE.parent / 'rfuse' / 'librfuse_hip.so'
SOURCES = ['capi.hip', 'conv3d.hip', 'conv3d_mfma.hip', 'conv3d_up.hip', 'conv3d_up_split.hip', 'conv3d_split.hip', 'conv3d_split_zc.hip', 'conv3d_e2_split.hip', 'conv3d_small.hip', 'conv3d_backward.hip', 'conv3d_wgrad_split.hip', 'conv_valid_mfma.hip', 'conv_valid_split.hip', 'conv_valid_split_pg.hip', 'linear.hip', 'attention.hip', 'attention_fused.hip', 'retrieval.hip', 'mesh.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
# the attention / gather-normalise kernels restate torch expressions op by op: no a*b+c fusion across operations (explicit fmaf() stays an FMA).
# hipcc's default -ffp-contract=fast fuses in the backend, where neither __fmul_rn nor `#pragma clang fp contract(off)` reach.
# conv3d_mfma.hip / conv3d_up.hip: no SLP vectorisation.  Their GroupNorm apply ((x - center) * scale + shift on register-staged halo rows) came out as
# v_pk_fma_f32 with scale and shift broadcast from ONE register pair by op_sel -- a packed-fp32 form that returns wrong results on gfx950 while another
# wave's F16 MFMA runs on the SIMD (check_isa below).  The scalar v_fma_f32 form gives the same bits and is immune.
EXTRA_FLAGS = {'attention.hip': ['-ffp-contract=off'], 'attention_fused.hip': ['-ffp-contract=off'], 'retrieval.hip': ['-ffp-contract=off'],
               'conv3d_mfma.hip': ['-fno-slp-vectorize'], 'conv3d_up.hip': ['-fno-slp-vectorize']}


def _hipcc():
    return os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def _llvm_bin():
    """llvm-objdump of the ROCm tree the compiler in use belongs to (HIPCC may point at another ROCm root than /opt/rocm)."""
    import shutil
    roots = [Path(os.path.realpath(_hipcc())).parents[1], Path(os.environ.get('ROCM_PATH', '/opt/rocm')), Path('/opt/rocm')]
    for r in roots:
        for sub in ('lib/llvm/bin', 'llvm/bin'):
            if (r / sub / 'llvm-objdump').exists():
                return r / sub
    found = shutil.which('llvm-objdump')
    if found:
        return Path(found).parent
    raise RuntimeError('check_isa: llvm-objdump not found under %s' % ', '.join(str(r) for r in roots))


def _stale(obj, deps):
    return (not obj.exists()) or any(d.stat().st_mtime > obj.stat().st_mtime for d in deps)


def unsafe_packed_fp32(asm_text):
    """Lines of a gfx950 disassembly that use a packed-fp32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, operands = 64-bit register
    pairs) with op_sel set for src1 or src2, i.e. whose LOW result lane takes the HIGH register of that pair.  Measured on MI355X
    (tools/pkfma_probe.py, profiles/r03_pkfma_probe.log): with op_sel[src1] = 1 -- or op_sel[src2] = 1 when src2 is the register pair of src1 -- lanes
    48..63 read that operand as ZERO whenever another wave's F16 MFMA executes on the same SIMD at that moment (never in a kernel running alone;
    op_sel on src0, op_sel_hi = 0 broadcasts, SGPR operands and plain forms are immune).  hipcc emits such forms freely (SLP-vectorised scalar code,
    `v.x + v.y` on a float2), so the build refuses any library that contains one."""
    import re
    bad = []
    for line in asm_text.splitlines():
        m = re.search(r'\bv_pk_(?:fma|mul|add)_f32\b.*?\bop_sel:\[([01,]+)\]', line)
        if m and '1' in m.group(1).split(',')[1:]:
            bad.append(line.strip())
    return bad


def check_isa(so=OUT, name=None):
    """Disassembles every gfx950 code object of the library and raises if an unsafe packed-fp32 form is in it (see unsafe_packed_fp32)."""
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        lib = Path(tmp) / (name or so.name)
        shutil.copy(so, lib)
        llvm_bin = _llvm_bin()
        subprocess.run([str(llvm_bin / 'llvm-objdump'), '--offloading', str(lib)], check=True, stdout=subprocess.DEVNULL, cwd=tmp)
        cos = sorted(Path(tmp).glob(lib.name + '.*gfx950*'))
        if not cos:
            raise RuntimeError('check_isa: no gfx950 code object found in %s' % so)
        bad = []
        for co in cos:
            asm = subprocess.run([str(llvm_bin / 'llvm-objdump'), '-d', str(co)], check=True, capture_output=True, text=True).stdout
            bad += unsafe_packed_fp32(asm)
    if bad:
        raise RuntimeError('check_isa: %d packed-fp32 instructions with op_sel on src1/src2 (wrong results beside F16 MFMAs on gfx950), e.g.\n  %s'
                           % (len(bad), '\n  '.join(bad[:5])))
    return len(cos)


def build(force=False, verbose=False, out=None, extra_flags=(), objdir=None):
    """``out`` / ``extra_flags`` / ``objdir``: a development variant of the library (e.g. -DRF_PERSIST_ROUNDS=2) built beside the product one, for same-box A/B
    runs through RFUSE_LIB (tools/ab_bench.sh); the product build takes none of them."""
    hipcc = _hipcc()
    OUT = Path(out) if out is not None else globals()['OUT']
    objdir = Path(objdir) if objdir is not None else HERE / 'build'
    objdir.mkdir(exist_ok=True, parents=True)
    headers = [HERE / 'common.h', HERE / 'conv_box.h', HERE / 'conv_split_common.h', HERE / 'attn_row.h', HERE.parents[1] / 'include' / 'rfuse.h']

    def compile_one(src):
        obj = objdir / (src.replace('.hip', '.o'))
        if force or _stale(obj, [HERE / src, HERE / 'build.py'] + headers):
            cmd = [hipcc] + FLAGS + list(extra_flags) + EXTRA_FLAGS.get(src, []) + ['-c', str(HERE / src), '-o', str(obj)]
            if verbose:
                print(' '.join(cmd))
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(OUT, objs):
        # link beside the target, check the ISA of THAT file, and only then move it into place: a library the check rejects (or that could not be
        # checked) never becomes rfuse/librfuse_hip.so, and a stale accepted one is removed rather than left to be loaded
        tmp_out = OUT.with_name(OUT.name + '.unchecked')
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(tmp_out)] + [str(o) for o in objs]
        if verbose:
            print(' '.join(cmd))
        try:
            subprocess.run(cmd, check=True)
            check_isa(tmp_out, name='librfuse_hip.so')
        except BaseException:
            tmp_out.unlink(missing_ok=True)
            OUT.unlink(missing_ok=True)
            raise
        os.replace(tmp_out, OUT)
    return OUT


TESTKIT_SRC = HERE.parents[1] / 'tests' / 'testkit' / 'testkit.hip'
TESTKIT_OUT = TESTKIT_SRC.parent / 'librfuse_testkit.so'


def build_testkit(force=False):
    """tests/testkit/librfuse_testkit.so: the test harness's own kernels (NaN poisoning of LDS / VGPRs, an F16-MFMA load).  Kept out of the product ABI."""
    if force or _stale(TESTKIT_OUT, [TESTKIT_SRC, HERE / 'build.py']):
        subprocess.run([_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', str(TESTKIT_OUT), str(TESTKIT_SRC)], check=True)
    return TESTKIT_OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
    print(build_testkit(force='--force' in sys.argv))
