"""model.loss (imported by the reference's trainers: trainer/train_refinement.py:13, trainer/train_retrieval.py:9) against
tests/golden/loss.npz, which oracle/gen_golden.py:gen_loss_fixture produced with the reference's own model/loss.py -- and the drop-in
package resolving EVERY name the reference's callers import from ``model`` (VERDICT r2 missing 1)."""
import importlib

import numpy as np
import pytest
import torch

from helpers import load_fixture


@pytest.fixture(scope='module')
def fix():
    return load_fixture('loss')


@pytest.mark.parametrize('cos', [True, False])
@pytest.mark.parametrize('temp', [0.5, 0.07])
def test_ntxent_matches_reference(fix, cos, temp):
    from model.loss import NTXentLoss
    key = '%s_t%g' % ('cos' if cos else 'dot', temp)
    crit = NTXentLoss(temp, cos)
    zis = torch.from_numpy(fix['zis']).requires_grad_(True)
    zjs = torch.from_numpy(fix['zjs']).requires_grad_(True)
    loss = crit(zis, zjs)
    loss.backward()
    assert loss.item() == pytest.approx(float(fix['loss_' + key]), rel=1e-6, abs=1e-7)
    np.testing.assert_allclose(zis.grad.numpy(), fix['grad_zis_' + key], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(zjs.grad.numpy(), fix['grad_zjs_' + key], rtol=1e-5, atol=1e-7)
    with torch.no_grad():
        liou = crit(torch.from_numpy(fix['zis']), torch.from_numpy(fix['zjs']), torch.from_numpy(fix['iou']))
    assert liou.item() == pytest.approx(float(fix['loss_iou_' + key]), rel=1e-6, abs=1e-7)


def test_style_and_normal_losses_match_reference(fix):
    from model.loss import patch_style_loss, get_cosine_similarity
    zis, zjs = torch.from_numpy(fix['zis']), torch.from_numpy(fix['zjs'])
    assert patch_style_loss(zis, zjs).item() == pytest.approx(float(fix['style']), rel=1e-6)
    got = get_cosine_similarity(torch.from_numpy(fix['pred_norms']), torch.from_numpy(fix['target_norms']))
    assert got.item() == pytest.approx(float(fix['normal_cos']), rel=1e-6)


def test_ntxent_is_device_agnostic():
    """the reference pins its mask with .cuda(device) (model/loss.py:57,62) and fails on CPU tensors; this one runs wherever the inputs are"""
    from model.loss import NTXentLoss
    z = torch.randn(4, 8)
    assert torch.isfinite(NTXentLoss(0.5, True)(z, z + 0.1 * torch.randn(4, 8)))


# every name the reference's callers import from the ``model`` package (trainer/train_refinement.py:11-13, trainer/train_retrieval.py:6,9,
# util/retrieval.py:14)
CALLER_IMPORTS = {
    'model': ['get_unet_backbone', 'get_decoder', 'get_retrieval_backbone', 'get_attention_block', 'get_retrieval_networks'],
    'model.attention': ['Unfold3D', 'Fold3D'],
    'model.loss': ['NTXentLoss', 'get_cosine_similarity'],
}


@pytest.mark.parametrize('module', sorted(CALLER_IMPORTS))
def test_every_name_the_reference_callers_import_resolves(module):
    mod = importlib.import_module(module)
    assert 'retrieval-fuse_amd' in mod.__file__, mod.__file__
    for name in CALLER_IMPORTS[module]:
        assert callable(getattr(mod, name)), (module, name)
