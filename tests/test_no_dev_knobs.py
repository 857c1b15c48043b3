"""The shipped library has no work-skipping developer switches: no RFUSE_* environment variable is read by the HIP sources
(RFUSE_CONV_ABLATE / _TILE / _MID / _CC8 lived in the shipping binary), and the tests / bench refuse to run with
one set (only RFUSE_LIB, the path of the shared library, is honoured)."""
import os
import re
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
ALLOWED = {'RFUSE_LIB'}


def test_kernel_sources_read_no_environment():
    for src in sorted((REPO / 'retrieval-fuse_amd' / 'csrc').glob('*.h*')):
        text = src.read_text()
        assert 'getenv' not in text, f'{src.name} reads the environment'


def test_python_side_reads_only_the_library_path():
    names = set()
    for src in list((REPO / 'retrieval-fuse_amd').rglob('*.py')) + [REPO / 'bench.py', REPO / '__graft_entry__.py']:
        names |= set(re.findall(r"RFUSE_[A-Z0-9_]+", src.read_text()))
    assert names <= ALLOWED | {'RFUSE_'}, names


def test_no_dev_knob_is_set_in_this_environment():
    bad = sorted(k for k in os.environ if k.startswith('RFUSE_') and k not in ALLOWED)
    assert not bad, f'developer switches set in the environment: {bad}'
