"""pytest wiring: registers the ``gpu`` marker and puts the product package + repo root on sys.path.

``-m "not gpu"`` runs here on CPU (oracle vs golden vectors, host logic, C-ABI symbol check, gloo world_size 2);
``-m gpu`` are the parity tests proper, run on a real MI355X through the C-ABI.
"""
import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
PKG = REPO / 'retrieval-fuse_amd'
for p in (str(REPO), str(PKG), str(REPO / 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return REPO / 'tests' / 'golden'


@pytest.fixture(scope='session')
def gpu():
    """cuda:0, or skip (test modules with their own ``gpu`` fixture shadow this one)"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    return torch.device('cuda:0')


@pytest.fixture(autouse=True)
def poison_gpu_memory(request):
    """Before every GPU test: fill the caching allocator's free memory with NaN bit patterns, so that `torch.empty` workspaces and
    outputs start as garbage, not as the zeros a fresh process usually sees -- a kernel that reads something it never wrote then
    fails every time instead of once in a blue moon (when the VRAM still holds another process's data).  The LDS of every CU and the
    vector register files get the same treatment (tests/testkit: rft_poison_lds / rft_poison_vgprs): neither is cleared between kernels."""
    if 'gpu' not in request.keywords:
        yield
        return
    import torch
    if torch.cuda.is_available():
        blocks = []
        try:
            for _ in range(6):                                          # 6 x 256 MiB: more than any test's working set of fresh blocks
                blocks.append(torch.full((64 * 1024 * 1024,), float('nan'), dtype=torch.float32, device='cuda:0'))
            for _ in range(256):                                        # the allocator's small-block pool (requests <= 1 MiB) is separate
                blocks.append(torch.full((256 * 1024 - 64,), float('nan'), dtype=torch.float32, device='cuda:0'))
        except RuntimeError:
            pass
        del blocks                                                      # back to the allocator's cache, contents intact
        import testkit
        tk = testkit.load()
        assert tk.rft_poison_lds(None) == 0                             # and NaNs in the LDS of every CU
        assert tk.rft_poison_vgprs(None) == 0                           # and in the vector register files (not cleared at wave launch)
        torch.cuda.synchronize()
    yield
