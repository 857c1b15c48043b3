"""ctypes loader of the test-harness kernels (tests/testkit/testkit.hip -> librfuse_testkit.so, built by retrieval-fuse_amd/csrc/build.py).
Not part of the product: nothing under retrieval-fuse_amd/ loads it."""
import ctypes
from pathlib import Path

import torch  # noqa: F401  (loads libamdhip64 first)

LIB_PATH = Path(__file__).resolve().parent / 'librfuse_testkit.so'
_lib = None


def load():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError('%s not found -- build it with `python retrieval-fuse_amd/csrc/build.py`' % LIB_PATH)
        _lib = ctypes.CDLL(str(LIB_PATH))
        _lib.rft_poison_lds.argtypes = [ctypes.c_void_p]
        _lib.rft_poison_vgprs.argtypes = [ctypes.c_void_p]
        _lib.rft_f16_mfma_load.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def f16_mfma_load(stream, out, blocks=256, iters=60000):
    """One 154-VGPR F16-MFMA workgroup per CU (blocks = 256), ~15 ms at iters = 60000, on `stream` (a torch.cuda.Stream)."""
    assert out.numel() >= blocks * 256 and out.dtype == torch.float32
    rc = load().rft_f16_mfma_load(blocks, iters, out.data_ptr(), stream.cuda_stream)
    if rc != 0:
        raise RuntimeError('rft_f16_mfma_load: HIP error %d' % rc)
