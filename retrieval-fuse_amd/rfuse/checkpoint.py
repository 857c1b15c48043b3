"""Checkpoint hand-over from the reference's trainers (SURVEY.md section 5 "Checkpoint / resume", Appendix B).

The reference saves PyTorch-Lightning checkpoints: ``{'state_dict': {'<attribute>.<module key>': tensor, ...}, ...}`` where ``<attribute>`` is the name
the LightningModule holds the network under -- ``unet_backbone``, ``decoder``, ``retrieval_backbone``, ``patched_attention_block``
(trainer/train_refinement.py:26-29) and ``fenc_input``, ``fenc_target`` (trainer/train_retrieval.py:23) -- and loads a network back with

    net.load_state_dict(rename_state_dict(ckpt['state_dict'], '<attribute>'))            util/misc.py:23-36, trainer/train_refinement.py:295-306

i.e. keep the keys that START WITH the attribute name and drop their first dotted component.  The modules here have the reference's state_dict keys,
order and shapes (tests/test_boundary.py), so the same rule loads the same files; ``load_state_dict`` stays strict: a missing or an unexpected key raises.
"""
from collections import OrderedDict
from pathlib import Path

REFINEMENT_PREFIXES = ('unet_backbone', 'decoder', 'retrieval_backbone', 'patched_attention_block')
RETRIEVAL_PREFIXES = ('fenc_input', 'fenc_target')


def rename_state_dict(state_dict, key):
    """util/misc.py:23-28, to the letter: ``k.startswith(key)`` (not ``key + '.'``), first dotted component dropped."""
    renamed = OrderedDict()
    for k in state_dict:
        if k.startswith(key):
            renamed['.'.join(k.split('.')[1:])] = state_dict[k]
    return renamed


def read_checkpoint(ckpt, map_location='cpu', trust_pickle=False):
    """a path (torch.load) or an already loaded checkpoint -> its ``state_dict`` mapping.  A bare state dict (no 'state_dict' entry) is taken as it is.
    Files are read with ``weights_only=True`` (tensors and plain containers only).  A Lightning checkpoint that pickles other objects beside its state dict
    (hyper-parameter namespaces, callbacks) needs ``trust_pickle=True``, which unpickles ARBITRARY code from the file: only for checkpoints you produced."""
    import torch
    if isinstance(ckpt, (str, Path)):
        try:
            ckpt = torch.load(str(ckpt), map_location=map_location, weights_only=True)
        except Exception as e:                                     # (pickle.UnpicklingError from the weights-only unpickler)
            if not trust_pickle:
                raise RuntimeError('%s holds more than tensors and plain containers (%s); pass trust_pickle=True to unpickle it -- that executes code from the '
                                   'file, so only for checkpoints whose origin you trust' % (ckpt, str(e).splitlines()[0][:200])) from e
            ckpt = torch.load(str(ckpt), map_location=map_location, weights_only=False)
    if not hasattr(ckpt, 'keys'):
        raise TypeError('a checkpoint is a path or a mapping, got %s' % type(ckpt).__name__)
    return ckpt['state_dict'] if 'state_dict' in ckpt else ckpt


def load_prefixed(modules, ckpt, prefixes, map_location='cpu', trust_pickle=False):
    """``modules``: {attribute name: nn.Module}.  Loads every module whose attribute name is in ``prefixes`` from ``ckpt`` by the reference's rule.
    A prefix without a single key in the checkpoint raises KeyError naming it (the reference would die inside load_state_dict with every key missing);
    missing / unexpected keys under a prefix raise RuntimeError from the strict ``load_state_dict``.  -> the attribute names loaded."""
    sd = read_checkpoint(ckpt, map_location, trust_pickle)
    loaded = []
    for name in prefixes:
        if name not in modules or modules[name] is None:
            continue
        part = rename_state_dict(sd, name)
        if not part:
            raise KeyError('checkpoint holds no %r.* keys (its prefixes: %s)' % (name, sorted({k.split('.')[0] for k in sd})))
        modules[name].load_state_dict(part)
        loaded.append(name)
    return loaded
