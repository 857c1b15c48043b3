"""CPU: the oracle (oracle/refpath.py) against the golden vectors captured from the reference (tests/golden/).
This is what pins the oracle; the GPU parity tests then compare the HIP path with the oracle / the same goldens."""
import contextlib
import io

import numpy as np
import pytest
import torch

import helpers
from oracle import refpath
from rfuse import configs as rf_configs
from rfuse import synthetic

NET_FIXTURES = ['net_C1', 'net_C2_stress_b2', 'net_C3', 'net_C4', 'net_C5']


def product_shapes(cfg):
    """state_dict key -> shape of the product modules (must equal the reference's, checked through the weight digest)."""
    import model
    with contextlib.redirect_stdout(io.StringIO()):
        mods = {'unet_backbone': model.get_unet_backbone(cfg), 'decoder': model.get_decoder(cfg),
                'retrieval_backbone': model.get_retrieval_backbone(cfg), 'patched_attention_block': model.get_attention_block(cfg)}
    return {k: {n: tuple(v.shape) for n, v in m.state_dict().items()} for k, m in mods.items()}


@pytest.mark.parametrize('name', NET_FIXTURES)
def test_oracle_forward_full_matches_reference(name):
    fix = helpers.load_fixture(name)
    cfg0 = rf_configs.get_config(str(fix['cfg_name']))
    cfg, x_in, retr, sds = helpers.fixture_problem(fix, product_shapes(cfg0))
    noise = torch.from_numpy(fix['gumbel_noise']) if 'gumbel_noise' in fix else None
    stages = {}
    torch.set_num_threads(8)
    with torch.no_grad():
        df = refpath.forward_full(sds, cfg, torch.from_numpy(x_in), torch.from_numpy(retr), float(fix['target_trunc']), noise, stages)
    # the oracle runs the same ATen kernels as the reference did, so agreement is essentially exact
    assert np.abs(df.numpy() - fix['df']).max() <= 1e-6
    assert np.abs(stages['x_back'][..., ::2, ::2, ::2].numpy() - fix['x_back_sub']).max() <= 1e-5
    assert np.abs(stages['x_retrieval'][..., ::4, ::4, ::4].numpy() - fix['x_retr_sub']).max() <= 1e-5
    assert np.abs(stages['x_attn'][..., ::2, ::2, ::2].numpy() - fix['x_attn_sub']).max() <= 1e-5
    for key, t in (('x_back_stats', stages['x_back']), ('x_retr_stats', stages['x_retrieval']), ('x_attn_stats', stages['x_attn']),
                   ('pred_stats', stages['pred'])):
        got = np.array([t.double().sum().item(), t.double().abs().sum().item(), (t.double() ** 2).sum().item()])
        np.testing.assert_allclose(got, fix[key], rtol=1e-6, atol=1e-4)


@pytest.mark.parametrize('name', ['net_C1', 'net_C4'])
def test_truth_fixture_is_the_oracle_in_float64(name):
    """The float64 'truth' the GPU parity bar is derived from (tests/golden/truth_*.npz) = this oracle evaluated in float64;
    for the softmax configs the generator also checked it against the reference's own modules in .double() (gap stored)."""
    fix, truth = helpers.load_truth(name)
    cfg0 = rf_configs.get_config(str(fix['cfg_name']))
    cfg, x_in, retr, sds = helpers.fixture_problem(fix, product_shapes(cfg0))
    sds64 = {m: {k: v.double() for k, v in sd.items()} for m, sd in sds.items()}
    noise = torch.from_numpy(fix['gumbel_noise']).double() if 'gumbel_noise' in fix else None
    torch.set_num_threads(8)
    with torch.no_grad():
        df = refpath.forward_full(sds64, cfg, torch.from_numpy(x_in).double(), torch.from_numpy(retr).double(), float(fix['target_trunc']), noise)
    assert np.abs(df.numpy() - truth['df_f64']).max() <= 1e-9
    prof = helpers.error_profile(fix['df'], truth['df_f64'])
    assert abs(prof['max'] - truth['ref_err_max']) <= 1e-9
    if not cfg['attn_retrieval_mode']:
        assert 0 <= truth['ref_double_gap'] <= 1e-9


@pytest.mark.parametrize('name', ['feat_C1', 'feat_C5'])
def test_oracle_get_features_matches_reference(name):
    """A8: PatchedAttentionBlock.get_features (model/attention.py:132-139) on x_back / the target's retrieval features."""
    ffix = helpers.load_fixture(name)
    nfix = helpers.load_fixture(name.replace('feat_', 'net_'))
    cfg0 = rf_configs.get_config(str(nfix['cfg_name']))
    cfg, x_in, retr, sds = helpers.fixture_problem(nfix, product_shapes(cfg0))
    seed = int(ffix['seed'])
    tgt = np.stack([synthetic.normalise_target(cfg, synthetic.make_chunk(seed * 100 + 90, cfg)['target_raw'])[None]])
    assert helpers.sha(tgt) == str(ffix['target_sha'])
    occ = torch.from_numpy(np.unpackbits(ffix['occupancy'])[:32 ** 3].astype(bool).reshape(1, 1, 32, 32, 32))
    torch.set_num_threads(8)
    with torch.no_grad():
        x_back = refpath.unet_backbone(torch.from_numpy(x_in), sds['unet_backbone'], cfg)
        x_target = refpath.fold3d(refpath.retrieval_backbone(refpath.unfold3d(torch.from_numpy(tgt), 16), sds['retrieval_backbone'], cfg), 4, 8, cfg['nf'])
        xf, pf, of = refpath.patched_get_features(x_back, x_target, occ, sds['patched_attention_block'], cfg)
    assert np.abs(x_target[..., ::4, ::4, ::4].numpy() - ffix['x_target_sub']).max() <= 1e-5
    assert np.abs(xf.numpy() - ffix['x_feat']).max() <= 1e-5 and np.abs(pf.numpy() - ffix['p_feat']).max() <= 1e-5
    np.testing.assert_array_equal(of.numpy(), np.unpackbits(ffix['occ_flat'])[:of.numel()].astype(bool))
    assert 0 < int(of.sum()) < of.numel()


def test_fold_unfold_inverse_and_order():
    x = torch.arange(2 * 3 * 8 * 8 * 8, dtype=torch.float32).reshape(2, 3, 8, 8, 8)
    rows = refpath.unfold3d(x, 2)
    assert rows.shape == (2 * 64, 3, 2, 2, 2)
    # row ((b*R+px)*R+py)*R+pz holds x[b, :, 2px:2px+2, 2py:2py+2, 2pz:2pz+2]   (model/attention.py:186-188)
    b, px, py, pz = 1, 2, 0, 3
    row = ((b * 4 + px) * 4 + py) * 4 + pz
    assert torch.equal(rows[row], x[b, :, 2 * px:2 * px + 2, 2 * py:2 * py + 2, 2 * pz:2 * pz + 2])
    assert torch.equal(refpath.fold3d(rows, 4, 2, 3), x)


@pytest.mark.parametrize('cfg_name', ['C1', 'C4', 'C5'])
def test_oracle_query_embedding_matches_reference(cfg_name):
    import model
    fix = helpers.load_fixture('query_' + cfg_name)
    cfg = rf_configs.get_config(cfg_name)
    trunc_i, _ = rf_configs.truncations(cfg)
    raw = synthetic.make_chunk(int(fix['seed']) * 100, cfg)['input_raw']
    windows = refpath.extract_query_windows(raw, cfg, trunc_i)
    assert tuple(fix['windows_shape']) == windows.shape
    assert helpers.sha(windows) == str(fix['windows_sha'])
    fenc_input, _ = model.get_retrieval_networks(cfg['retrieval_model'])
    shapes = {k: tuple(v.shape) for k, v in fenc_input.state_dict().items()}
    sd = helpers.seeded_sd(shapes, int(fix['seed']) * 1000 + helpers.SD_OFFSETS['fenc_input'])
    with torch.no_grad():
        emb = refpath.embed_queries(windows, sd, cfg)
    assert np.abs(emb.numpy() - fix['emb']).max() <= 1e-6


def test_oracle_knn_demotion_compose_match_reference():
    fix = helpers.load_fixture('retrieval_map_compose')
    cfg = rf_configs.get_config('C1')
    _, trunc_t = rf_configs.truncations(cfg)
    seed, n_patches, q_scene = int(fix['seed']), int(fix['n_patches']), int(fix['q_scene'])
    db = synthetic.make_database(seed, cfg, n_patches)
    assert helpers.sha(db['meta'], db['emb'], db['volumes']) == str(fix['db_sha'])
    K = cfg['K']
    idx, dist = refpath.knn_exact(fix['queries'], db['emb'], 2 * K)
    rows = refpath.mapping_rows(idx, dist, db['meta'])
    m_train = refpath.demote_same_scene(rows, np.full(64, q_scene), K)
    m_val = refpath.demote_same_scene(rows, np.full(64, -1), K)
    # indices / extents exact; distances to float32 rounding of the float64 brute force
    np.testing.assert_array_equal(m_train[..., :7], fix['map_train'][..., :7])
    np.testing.assert_array_equal(m_val[..., :7], fix['map_val'][..., :7])
    np.testing.assert_allclose(m_train[..., 7], fix['map_train'][..., 7], rtol=0, atol=1e-6)
    assert (m_train[..., :7] != m_val[..., :7]).any(), 'fixture must exercise the demotion'
    c_train = refpath.compose_retrieval(fix['map_train'], db['volumes'], K, trunc_t)
    c_val = refpath.compose_retrieval(fix['map_val_sentinel'], db['volumes'], K, trunc_t)
    assert helpers.sha(c_train) == str(fix['compose_train_sha'])
    assert helpers.sha(c_val) == str(fix['compose_val_sha'])
    # query-side occupancy filter: dropped patches keep the trunc initialisation (util/retrieval.py:148,151)
    keep = fix['patch_keep']
    assert 0 < keep.sum() < 64
    c_masked = refpath.compose_retrieval(fix['map_val_sentinel'], db['volumes'], K, trunc_t, patch_keep=keep)
    assert helpers.sha(c_masked) == str(fix['compose_masked_sha'])
    assert (c_masked != c_val).any()
    # the reference's own extent enumeration (dataset/scene.py:152-160) vs the product's patch boxes
    ext = fix['extents_64_16_8_16']
    boxes = synthetic.patch_boxes_64()
    np.testing.assert_array_equal(ext[:, [0, 2, 4]], boxes[:, [0, 2, 4]])
    np.testing.assert_array_equal(ext[:, [1, 3, 5]] - 16, boxes[:, [1, 3, 5]])


def test_oracle_overlapping_compose_matches_reference():
    """create_retrieval_from_mapping with dataset.no_overlap False (util/retrieval.py:156), an order-dependent reduction: the restatement against the reference's
    output on a 7^3 grid of 16^3 patches at stride 8 (tests/golden/compose_overlap.npz, generated by oracle/gen_golden.py from the reference itself)"""
    fix = helpers.load_fixture('compose_overlap')
    cfg = rf_configs.get_config('C1')
    _, trunc_t = rf_configs.truncations(cfg)
    db = synthetic.make_database(int(fix['seed']), cfg, int(fix['n_db_patches']))
    assert helpers.sha(db['meta'], db['emb'], db['volumes']) == str(fix['db_sha'])
    got = refpath.compose_retrieval_overlap(fix['mapping'], fix['boxes'], db['volumes'], cfg['K'], trunc_t)
    assert helpers.sha(got) == str(fix['composed_sha'])
    np.testing.assert_array_equal(got[:, ::4, ::4, ::4], fix['composed_sub'])
    assert float(fix['min_relative_margin']) > 1e-5 and 0 < int(fix['taken']) < int(fix['visited'])
    # with no_overlap the gate is open for every patch: later patches always win
    always = refpath.compose_retrieval_overlap(fix['mapping'], fix['boxes'], db['volumes'], cfg['K'], trunc_t, no_overlap=True)
    assert (always != got).any()
    # boxes = the padded extents without their context (dataset/patched_scene_dataset.py:103-107)
    np.testing.assert_array_equal(fix['boxes'][:, 0::2], fix['extents'][:, 0::2])
    np.testing.assert_array_equal(fix['boxes'][:, 1::2], fix['extents'][:, 1::2] - 16)


def test_knn_ties_go_to_lower_index():
    db = np.zeros((5, 64), dtype=np.float32)
    db[3, 0] = 1.0
    q = np.zeros((1, 64), dtype=np.float32)
    idx, dist = refpath.knn_exact(q, db, 4)
    assert idx.tolist() == [[0, 1, 2, 4]] and dist[0, 0] == 0.0
