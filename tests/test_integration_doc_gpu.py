"""INTEGRATION.md §3 shows the ctypes stub a maintainer of the reference would write.  Run that block AS WRITTEN (extracted from
the markdown) so the document cannot drift from include/rfuse.h: SingleConv 'gcr' (reference model/unet.py:19-100) against
torch.nn.functional in float64."""
import os
import re

import pytest
import torch
import torch.nn.functional as F

from helpers import GOLDEN

REPO = str(GOLDEN.parents[1])

pytestmark = pytest.mark.gpu


def stub_source():
    text = open(os.path.join(REPO, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    hits = [b for b in blocks if 'def single_conv_gcr' in b]
    assert len(hits) == 1
    return hits[0]


@pytest.mark.parametrize('n,c,cout,s,groups', [(3, 16, 32, 8, 8), (2, 5, 16, 4, 8)])
def test_the_ctypes_stub_of_the_integration_guide_runs_as_written(n, c, cout, s, groups):
    src = stub_source().replace("'retrieval-fuse_amd/rfuse/librfuse_hip.so'",
                                repr(os.path.join(REPO, 'retrieval-fuse_amd', 'rfuse', 'librfuse_hip.so')))
    ns = {}
    exec(compile(src, 'INTEGRATION.md#3', 'exec'), ns)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, c, s, s, s, generator=g).cuda()
    gamma = (1 + 0.3 * torch.randn(c, generator=g)).cuda()
    beta = (0.2 * torch.randn(c, generator=g)).cuda()
    w = (torch.randn(cout, c, 3, 3, 3, generator=g) / (27 * c) ** 0.5).cuda()
    got = ns['single_conv_gcr'](x, gamma, beta, w, groups)
    torch.cuda.synchronize()
    geff = groups if c >= groups else 1
    want = F.relu(F.conv3d(F.group_norm(x.double(), geff, gamma.double(), beta.double(), 1e-5), w.double(), padding=1))
    assert (got.double() - want).abs().max().item() < 2e-5
