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
for p in (str(REPO), str(PKG)):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return REPO / 'tests' / 'golden'
