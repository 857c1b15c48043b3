"""Checkpoint hand-over: a Lightning-shaped ``{'state_dict': {'<attribute>.<key>': tensor}}`` written with torch.save
loads into the drop-in modules by the reference's prefix rule (util/misc.py:23-28; trainer/train_refinement.py:295-306; util/retrieval.py:224-225)."""
import contextlib
import io

import numpy as np
import pytest
import torch

import helpers
import model
from rfuse import checkpoint, configs as rf_configs

PREFIXES = ('unet_backbone', 'decoder', 'retrieval_backbone', 'patched_attention_block', 'fenc_input', 'fenc_target')


def _modules(cfg):
    with contextlib.redirect_stdout(io.StringIO()):
        fenc_input, fenc_target = model.get_retrieval_networks(cfg['retrieval_model'])
        return {'unet_backbone': model.get_unet_backbone(cfg), 'decoder': model.get_decoder(cfg), 'retrieval_backbone': model.get_retrieval_backbone(cfg),
                'patched_attention_block': model.get_attention_block(cfg), 'fenc_input': fenc_input, 'fenc_target': fenc_target}


def _lightning_checkpoints(mods, seed):
    """two files' worth, as the reference's trainers write them: refinement (four attributes + keys of things that are not networks) and retrieval"""
    sds = {n: helpers.seeded_sd({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed + i) for i, (n, m) in enumerate(mods.items())}
    flat = lambda names: {'%s.%s' % (n, k): v for n in names for k, v in sds[n].items()}
    refinement = {'epoch': 7, 'global_step': 1234, 'state_dict': dict(flat(PREFIXES[:4]), **{'loss_ntxent.temperature': torch.ones(1)}), 'hyper_parameters': {'K': 4}}
    retrieval = {'epoch': 3, 'state_dict': flat(PREFIXES[4:])}
    return sds, refinement, retrieval


@pytest.mark.parametrize('cfg_name', ['C1', 'C5'])
def test_lightning_checkpoints_load_by_the_reference_prefix_rule(tmp_path, cfg_name):
    cfg = rf_configs.get_config(cfg_name)
    mods = _modules(cfg)
    sds, refinement, retrieval = _lightning_checkpoints(mods, 900)
    torch.save(refinement, tmp_path / 'refinement.ckpt')
    torch.save(retrieval, tmp_path / 'retrieval.ckpt')
    fresh = _modules(cfg)
    assert checkpoint.load_prefixed(fresh, tmp_path / 'refinement.ckpt', checkpoint.REFINEMENT_PREFIXES) == list(PREFIXES[:4])
    assert checkpoint.load_prefixed(fresh, str(tmp_path / 'retrieval.ckpt'), checkpoint.RETRIEVAL_PREFIXES) == list(PREFIXES[4:])
    for name in PREFIXES:
        got = fresh[name].state_dict()
        assert list(got) == list(sds[name])                              # every tensor, in the reference's key order
        for k in got:
            assert torch.equal(got[k], sds[name][k]), (name, k)
    # the rule itself, to the letter: startswith(key), first dotted component dropped
    r = checkpoint.rename_state_dict({'decoder.network.1.bias': 1, 'decoderX.y.z': 2, 'unet_backbone.a': 3}, 'decoder')
    assert dict(r) == {'network.1.bias': 1, 'y.z': 2}


def test_missing_and_unexpected_keys_raise(tmp_path):
    cfg = rf_configs.get_config('C1')
    mods = _modules(cfg)
    _, refinement, retrieval = _lightning_checkpoints(mods, 50)
    broken = dict(refinement, state_dict=dict(refinement['state_dict']))
    victim = next(k for k in broken['state_dict'] if k.startswith('retrieval_backbone.') and k.endswith('conv.weight'))
    del broken['state_dict'][victim]
    with pytest.raises(RuntimeError, match='Missing key'):
        checkpoint.load_prefixed(_modules(cfg), broken, checkpoint.REFINEMENT_PREFIXES)
    extra = dict(refinement, state_dict=dict(refinement['state_dict'], **{'decoder.network.9.weight': torch.zeros(1)}))
    with pytest.raises(RuntimeError, match='Unexpected key'):
        checkpoint.load_prefixed(_modules(cfg), extra, checkpoint.REFINEMENT_PREFIXES)
    with pytest.raises(KeyError, match='fenc_input'):                     # the refinement file handed in as the retrieval checkpoint
        checkpoint.load_prefixed(_modules(cfg), refinement, checkpoint.RETRIEVAL_PREFIXES)
    wrong_shape = dict(retrieval, state_dict={k: (v[:1] if k == 'fenc_input.layers.0.bias' else v) for k, v in retrieval['state_dict'].items()})
    with pytest.raises(RuntimeError, match='size mismatch'):
        checkpoint.load_prefixed(_modules(cfg), wrong_shape, checkpoint.RETRIEVAL_PREFIXES)
    with pytest.raises(TypeError):
        checkpoint.read_checkpoint(3)


@pytest.mark.gpu
def test_engine_load_checkpoints(gpu, tmp_path):
    """RefinementEngine.load_checkpoints(refinement.ckpt, retrieval.ckpt) == load_state_dicts of the same tensors: identical df"""
    from rfuse import synthetic
    from rfuse.database import PatchDatabase
    from rfuse.engine import RefinementEngine
    cfg = rf_configs.get_config('C3')
    db = synthetic.make_database(4, cfg, 640)
    eng = RefinementEngine(cfg, gpu, PatchDatabase(db['emb'], db['meta'], db['volumes'], gpu))
    mods = dict(eng.modules(), fenc_target=eng.fenc_target)
    sds, refinement, retrieval = _lightning_checkpoints(mods, 400)
    torch.save(refinement, tmp_path / 'refinement.ckpt')
    torch.save(retrieval, tmp_path / 'retrieval.ckpt')
    raw = torch.from_numpy(np.stack([synthetic.make_chunk(31 + b, cfg)['input_raw'] for b in range(2)])).to(gpu)
    eng.load_state_dicts({n: sds[n] for n in eng.modules()})
    want = eng.refine(raw).clone()
    other = RefinementEngine(cfg, gpu, eng.database)
    assert sorted(other.load_checkpoints(tmp_path / 'refinement.ckpt', tmp_path / 'retrieval.ckpt')) == sorted(PREFIXES)
    assert torch.equal(other.refine(raw), want)
    assert all(torch.equal(v, sds['fenc_target'][k].to(gpu)) for k, v in other.fenc_target.state_dict().items())
