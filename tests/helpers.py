"""Shared test helpers: rebuild the exact inputs / weights a golden fixture was generated from (seeded,
digest-checked), and load fixtures.  Mirrors oracle/gen_golden.py's recipe without touching /root/reference."""
import hashlib
from pathlib import Path

import numpy as np
import torch

from rfuse import configs as rf_configs
from rfuse import synthetic

GOLDEN = Path(__file__).resolve().parent / 'golden'

# state_dict seeds used by oracle/gen_golden.py: seed*1000 + offset
SD_OFFSETS = {'unet_backbone': 11, 'decoder': 12, 'retrieval_backbone': 13, 'patched_attention_block': 14, 'fenc_input': 21}


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def load_fixture(name):
    z = np.load(GOLDEN / (name + '.npz'), allow_pickle=False)
    return {k: z[k] for k in z.files}


def load_truth(name):
    """-> (fixture, truth) where truth holds the float64 evaluation of the same network on the same inputs
    (oracle/gen_golden.py:gen_truth_fixture): ``df_f64`` [B,1,64^3] and the sub-sampled stages, rebuilt from the float32
    residuals stored against the reference's fp32 results."""
    fix = load_fixture(name)
    t = load_fixture(name.replace('net_', 'truth_'))
    truth = {'df_f64': fix['df'].astype(np.float64) + t['df_resid'].astype(np.float64)}
    for k in ('x_back', 'x_retr', 'x_attn'):
        truth[k + '_f64_sub'] = fix[k + '_sub'].astype(np.float64) + t[k + '_resid_sub'].astype(np.float64)
    truth.update({k: float(t[k]) for k in ('ref_err_max', 'ref_err_rms', 'ref_frac_gt_1e4', 'ref_double_gap')})
    return fix, truth


def error_profile(a, truth):
    """distribution of |a - truth|: max, rms, selected quantiles and the fraction above north_star's 1e-4"""
    e = np.abs(np.asarray(a, dtype=np.float64) - truth).ravel()
    q = np.quantile(e, [0.5, 0.99, 0.999, 0.9999])
    return {'max': float(e.max()), 'rms': float(np.sqrt((e ** 2).mean())), 'p50': float(q[0]), 'p99': float(q[1]), 'p99.9': float(q[2]),
            'p99.99': float(q[3]), 'frac>1e-4': float((e > 1e-4).mean())}


def chunk_inputs(cfg, seed, batch, stress=False):
    trunc_i, trunc_t = rf_configs.truncations(cfg)
    xs, rs = [], []
    for b in range(batch):
        if stress and cfg['task'] != 'surface_reconstruction':
            s_in = cfg['dataset_train']['input_chunk_size']
            raw_in = synthetic.uniform_stress_volume(seed * 100 + b, (s_in,) * 3, trunc_i)
            raw_r = [synthetic.uniform_stress_volume(seed * 100 + 50 + b * 10 + k, (64,) * 3, trunc_t) for k in range(cfg['K'])]
        else:
            raw_in = synthetic.make_chunk(seed * 100 + b, cfg)['input_raw']
            raw_r = [synthetic.make_chunk(seed * 100 + 50 + b * 10 + k, cfg)['target_raw'] for k in range(cfg['K'])]
        xs.append(synthetic.normalise_input(cfg, raw_in)[None])
        rs.append(np.stack([synthetic.normalise_target(cfg, r) for r in raw_r]))
    return np.stack(xs).astype(np.float32), np.stack(rs).astype(np.float32)


def seeded_sd(shapes, seed, as_torch=True):
    sd = synthetic.seeded_state_dict(shapes, seed)
    return {k: torch.from_numpy(v) for k, v in sd.items()} if as_torch else sd


def fixture_problem(fix, shapes_by_module):
    """Rebuild (cfg, x_in, retr, state_dicts) for a network fixture and verify the digests stored in it.
    ``shapes_by_module``: {'unet_backbone': {key: shape}, ...} -- taken from the product modules' state_dict()."""
    cfg = rf_configs.get_config(str(fix['cfg_name']))
    seed, batch, stress = int(fix['seed']), int(fix['batch']), bool(int(fix['stress']))
    x_in, retr = chunk_inputs(cfg, seed, batch, stress)
    assert sha(x_in, retr) == str(fix['inputs_sha']), 'synthetic generator drifted from the golden fixture'
    sds = {m: seeded_sd(shapes_by_module[m], seed * 1000 + SD_OFFSETS[m]) for m in
           ('unet_backbone', 'decoder', 'retrieval_backbone', 'patched_attention_block')}
    digest = sha(*[v.numpy() for m in ('unet_backbone', 'decoder', 'retrieval_backbone', 'patched_attention_block') for v in sds[m].values()])
    assert digest == str(fix['weights_sha']), 'state_dict key order / shapes differ from the reference modules'
    return cfg, x_in, retr, sds
