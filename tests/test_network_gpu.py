"""GPU parity of the whole refinement path: the drop-in modules (model.*) and the online engine against the golden
vectors captured from the reference (tests/golden/) and against the oracle.  Bar (BASELINE.json north_star):
<= 1e-4 abs on the reconstructed TSDF ``df``; top-k indices exact (checked in test_kernels_gpu.py)."""
import contextlib
import io

import numpy as np
import pytest
import torch

import helpers
from oracle import refpath
from rfuse import configs as rf_configs
from rfuse import synthetic

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
DF_TOL = 1e-4          # abs, on reconstructed TSDF values (north_star)


def df_bar(name):
    """The parity bar on ``df`` for a golden fixture, DERIVED FROM COMMITTED DATA instead of chosen by hand.

    north_star asks for <= 1e-4 abs against the reference.  The reference result is itself an fp32 evaluation; the committed
    truth_*.npz fixtures hold the float64 evaluation of the same network on the same inputs, so the distance of the
    reference's own arithmetic from the truth is known per fixture.  Where that distance is below 1e-4 (ShapeNetV2, 3DFront:
    <= 5e-6) the bar is north_star's 1e-4 against the reference.  Where it is NOT (Matterport3D: distances in units 180x
    larger, trunc 11.25, softmax sharpness 1024 -- the reference sits 4.1e-4 from the truth, 0.17 % of the voxels above
    1e-4), "within 1e-4 of the reference" is not a property an exact evaluation would have; there the HIP path must be no
    further from the truth than the reference is:  max|gpu - f64| <= 1.25 max|ref - f64|  and
    frac(|gpu - ref| > 1e-4) <= 2 frac(|ref - f64| > 1e-4)."""
    fix, truth = helpers.load_truth(name)
    ref_prof = helpers.error_profile(fix['df'], truth['df_f64'])
    return fix, truth, ref_prof, ref_prof['max'] <= 0.1 * DF_TOL


def assert_df_parity(name, df_gpu, label='df'):
    fix, truth, ref_prof, plain_bar = df_bar(name)
    gpu_prof = helpers.error_profile(df_gpu, truth['df_f64'])
    vs_ref = helpers.error_profile(df_gpu, fix['df'].astype(np.float64))
    fmt = lambda p: '  '.join('%s %.2e' % (k, v) for k, v in p.items())
    print(f'\n{name} {label}\n   ref vs f64: {fmt(ref_prof)}\n   gpu vs f64: {fmt(gpu_prof)}\n   gpu vs ref: {fmt(vs_ref)}')
    # north_star's 1e-4 on the network's OWN output, whatever the truncation: pred = tanh(...) in (-1, 1) and df = (pred + 1) * trunc / 2
    # (trainer/train_refinement.py:242-243), so |pred_gpu - pred_ref| = |df_gpu - df_ref| * 2 / trunc -- the form of the bar that CAN be met on Matterport3D
    # (trunc 11.25: a df error of 3.5e-4 is 6e-5 of the tanh output)
    trunc = float(fix['target_trunc'])
    pred_err = vs_ref['max'] * 2.0 / trunc
    print(f'   tanh output: max |pred_gpu - pred_ref| = {pred_err:.2e} (trunc {trunc})')
    if trunc > 1.0:
        assert pred_err <= DF_TOL, f'{label}: network output (tanh) differs from the reference by {pred_err:.3e} > 1e-4'
    if plain_bar:
        assert vs_ref['max'] <= DF_TOL, f'{label} max abs err vs reference {vs_ref["max"]:.3e}'
    else:
        assert gpu_prof['max'] <= 1.25 * ref_prof['max'], f'gpu is {gpu_prof["max"]:.3e} from the float64 truth, the reference {ref_prof["max"]:.3e}'
        assert vs_ref['frac>1e-4'] <= 2 * ref_prof['frac>1e-4'], f'{vs_ref["frac>1e-4"]:.5f} of the voxels differ from the reference by > 1e-4'
        assert gpu_prof['rms'] <= 1.25 * ref_prof['rms']
    return gpu_prof, vs_ref


def df_tolerance(trunc):
    """engine-vs-oracle comparisons on problems that have no float64 truth fixture: 1e-4 abs (north_star) where the truncation
    is O(0.1) scene units (ShapeNetV2 0.0625, 3DFront 0.1625); for Matterport3D (trunc 11.25) twice the reference's own
    measured distance from the truth on the committed C4 fixture (truth_C4.npz: ref_err_max) -- both sides of such a
    comparison are fp32 evaluations."""
    if trunc <= 1.0:
        return DF_TOL
    return 2.0 * float(helpers.load_fixture('truth_C4')['ref_err_max'])


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    return torch.device(DEV)


def build_modules(cfg):
    import model
    with contextlib.redirect_stdout(io.StringIO()):
        return {'unet_backbone': model.get_unet_backbone(cfg), 'decoder': model.get_decoder(cfg),
                'retrieval_backbone': model.get_retrieval_backbone(cfg), 'patched_attention_block': model.get_attention_block(cfg)}


def maxerr(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


@pytest.mark.parametrize('name', ['net_C1', 'net_C2_stress_b2', 'net_C3', 'net_C4', 'net_C5'])
def test_modules_match_reference_golden(gpu, name):
    """forward_full wiring with the drop-in classes exactly as the reference trainer uses them
    (trainer/train_refinement.py:108-116): Unfold3D(16,1) -> retrieval_backbone -> Fold3D(4,8,nf) -> attention -> decoder."""
    from model.attention import Unfold3D, Fold3D
    fix = helpers.load_fixture(name)
    cfg0 = rf_configs.get_config(str(fix['cfg_name']))
    mods = build_modules(cfg0)
    shapes = {k: {n: tuple(v.shape) for n, v in m.state_dict().items()} for k, m in mods.items()}
    cfg, x_in, retr, sds = helpers.fixture_problem(fix, shapes)
    for k, m in mods.items():
        m.load_state_dict(sds[k])
        m.to(gpu).eval()
    K, B = cfg['K'], x_in.shape[0]
    trunc_t = float(fix['target_trunc'])
    noise = torch.from_numpy(fix['gumbel_noise']).to(gpu) if 'gumbel_noise' in fix else None
    with torch.no_grad():
        x_back = mods['unet_backbone'](torch.from_numpy(x_in).to(gpu))
        retrievals = torch.from_numpy(retr).to(gpu)[:, :K].reshape(B * K, 1, 64, 64, 64)
        feats = mods['retrieval_backbone'](Unfold3D(16, 1)(retrievals))
        x_retr = Fold3D(4, 8, cfg['nf'])(feats)
        x_attn = mods['patched_attention_block'](x_back, x_retr, noise)
        pred = mods['decoder'](x_attn)
        df = (pred + 1) * trunc_t / 2
        # the fused route: patch-major attention input + df epilogue
        x_attn2 = mods['patched_attention_block'].forward_patch_major(x_back, feats, 8, noise)
        df2 = mods['decoder'].forward_df(x_attn2, trunc_t)
    e_back = maxerr(x_back[..., ::2, ::2, ::2].cpu(), fix['x_back_sub'])
    e_retr = maxerr(x_retr[..., ::4, ::4, ::4].cpu(), fix['x_retr_sub'])
    e_attn = maxerr(x_attn[..., ::2, ::2, ::2].cpu(), fix['x_attn_sub'])
    e_df = maxerr(df.cpu(), fix['df'])
    print(f'\n{name}: x_back {e_back:.2e}  x_retr {e_retr:.2e}  x_attn {e_attn:.2e}  df {e_df:.2e}  (trunc {trunc_t})')
    # intermediates: fp32 round-off chained through ~20 GroupNorm+conv layers (per-layer error is ~1e-6, tools/debug_layers.py)
    assert e_back <= 1e-4 * max(1.0, float(np.abs(fix['x_back_sub']).max()))
    assert e_retr <= 1e-4 * max(1.0, float(np.abs(fix['x_retr_sub']).max()))
    assert torch.equal(x_attn, x_attn2) and torch.equal(df, df2), 'patch-major route must equal the folded route bit for bit'
    assert_df_parity(name, df.cpu().numpy())
    # the stages against the float64 truth, next to the reference's own distance from it
    _, truth = helpers.load_truth(name)
    t = helpers.load_fixture(name.replace('net_', 'truth_'))
    for tag, got, key, stride in (('x_back', x_back, 'ref_err_back', 2), ('x_retr', x_retr, 'ref_err_retr', 4), ('x_attn', x_attn, 'ref_err_attn', 2)):
        g = maxerr(got[..., ::stride, ::stride, ::stride].cpu(), truth[tag + '_f64_sub'])
        print(f'   {tag}: gpu vs f64 {g:.2e} (sub-sampled)   ref vs f64 {float(t[key]):.2e} (full)')


@pytest.mark.parametrize('cfg_name', ['C1', 'C4', 'C5'])
def test_query_encoder_matches_reference_golden(gpu, cfg_name):
    import model
    fix = helpers.load_fixture('query_' + cfg_name)
    cfg = rf_configs.get_config(cfg_name)
    trunc_i, _ = rf_configs.truncations(cfg)
    raw = synthetic.make_chunk(int(fix['seed']) * 100, cfg)['input_raw']
    fenc_input, _ = model.get_retrieval_networks(cfg['retrieval_model'])
    shapes = {k: tuple(v.shape) for k, v in fenc_input.state_dict().items()}
    fenc_input.load_state_dict(helpers.seeded_sd(shapes, int(fix['seed']) * 1000 + helpers.SD_OFFSETS['fenc_input']))
    from rfuse.engine import RefinementEngine
    eng = RefinementEngine(cfg, gpu)
    eng.fenc_input.load_state_dict(fenc_input.state_dict())
    emb = eng.embed_queries(torch.from_numpy(raw[None]).to(gpu))
    assert maxerr(emb.cpu(), fix['emb']) <= 2e-6


def test_engine_online_path_matches_oracle(gpu):
    """retrieve (embed -> exact kNN -> demotion -> gather) + attend + refine, end to end, vs the oracle doing the same
    with float64 kNN and the reference's compose semantics."""
    from rfuse.database import PatchDatabase
    from rfuse.engine import RefinementEngine
    cfg = rf_configs.get_config('C3')                      # softmax attention: deterministic
    trunc_i, trunc_t = rf_configs.truncations(cfg)
    K, B = cfg['K'], 2
    db = synthetic.make_database(21, cfg, 64 * 40)
    eng = RefinementEngine(cfg, gpu, PatchDatabase(db['emb'], db['meta'], db['volumes'], gpu))
    sds = {}
    for name, m in eng.modules().items():
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        sds[name] = helpers.seeded_sd(shapes, 900 + len(name))
    eng.load_state_dicts(sds)
    raws = np.stack([synthetic.make_chunk(300 + b, cfg)['input_raw'] for b in range(B)])
    qscene = np.concatenate([np.full(64, 2), np.full(64, -1)]).astype(np.int32)
    df = eng.refine(torch.from_numpy(raws).to(gpu), torch.from_numpy(qscene).to(gpu)).cpu().numpy()

    # ---- oracle
    with torch.no_grad():
        q = torch.cat([refpath.embed_queries(refpath.extract_query_windows(r, cfg, trunc_i), sds['fenc_input'], cfg) for r in raws]).numpy()
    idx, dist = refpath.knn_exact(q, db['emb'], 2 * K)
    mapping = refpath.demote_same_scene(refpath.mapping_rows(idx, dist, db['meta']), qscene, K)
    d = cfg['dataset_train']
    retr = np.stack([refpath.compose_retrieval(mapping[b * 64:(b + 1) * 64], db['volumes'], K, trunc_t) for b in range(B)])
    retr = ((retr - np.float32(d['target_mean'])) / np.float32(d['target_std'])).astype(np.float32)
    x_in = np.stack([synthetic.normalise_input(cfg, r)[None] for r in raws])
    with torch.no_grad():
        df_ref = refpath.forward_full(sds, cfg, torch.from_numpy(x_in), torch.from_numpy(retr), trunc_t).numpy()
    # the engine's retrieval must pick the same database rows (exact kNN), then the fields agree to tolerance
    patches, meta = eng.retrieve(torch.from_numpy(raws).to(gpu), torch.from_numpy(qscene).to(gpu))
    np.testing.assert_array_equal(meta.cpu().numpy(), mapping[..., :7].astype(np.int32))
    assert maxerr(df, df_ref) <= df_tolerance(trunc_t)


def test_state_dict_roundtrip_and_repack(gpu):
    """load_state_dict after a forward must invalidate the packed-weight cache."""
    cfg = rf_configs.get_config('C1')
    mods = build_modules(cfg)
    dec = mods['decoder'].to(gpu).eval()
    x = torch.rand(1, 16, 32, 32, 32, device=gpu)
    with torch.no_grad():
        a = dec(x)
        sd = {k: v * 0.5 if 'conv.weight' in k else v for k, v in dec.state_dict().items()}
        dec.load_state_dict(sd)
        b = dec(x)
    assert not torch.equal(a, b)


def test_feature_cache_mode_matches_full_path(gpu):
    """optional serving mode: cached retrieval-backbone features per database row == recomputing them per query"""
    from rfuse.database import PatchDatabase
    from rfuse.engine import RefinementEngine
    cfg = rf_configs.get_config('C3')
    db = synthetic.make_database(33, cfg, 64 * 6 + 10)
    eng = RefinementEngine(cfg, gpu, PatchDatabase(db['emb'], db['meta'], db['volumes'], gpu))
    sds = {}
    for name, m in eng.modules().items():
        sds[name] = helpers.seeded_sd({k: tuple(v.shape) for k, v in m.state_dict().items()}, 400 + len(name))
    eng.load_state_dicts(sds)
    # make the sentinel row a likely hit for one query chunk: query == sentinel embedding direction is random, so force it
    raws = np.stack([synthetic.make_chunk(700 + b, cfg)['input_raw'] for b in range(2)])
    x = torch.from_numpy(raws).to(gpu)
    full = eng.refine(x)
    with pytest.raises(RuntimeError, match='build_feature_cache'):
        eng.refine(x, use_feature_cache=True)
    cache = eng.database.build_feature_cache(eng.retrieval_backbone, cfg, rows_per_batch=128)
    assert tuple(cache.shape) == (64 * 6 + 11, cfg['nf'], 8, 8, 8)
    cached = eng.refine(x, use_feature_cache=True)
    assert maxerr(cached.cpu(), full.cpu()) <= 1e-5


@pytest.mark.parametrize('cfg_name,B', [('C1', 8), ('C3', 8), ('C4', 4), ('C5', 4)])
def test_engine_fast_routes_match_plain_routes_at_bench_sizes(gpu, cfg_name, B):
    """At batch sizes where the big-tile kernels engage (parity-split decoder conv, fused max-pool / pooled-only encoder
    levels, volume-domain attention with the fused MLP), the engine must give the same field as the plain routes
    (generic conv on the virtually upsampled source, stand-alone max-pool, per-layer linear + row-domain attention)."""
    from rfuse import ops
    from rfuse.database import PatchDatabase
    from rfuse.engine import RefinementEngine
    cfg = rf_configs.get_config(cfg_name)
    _, trunc_t = rf_configs.truncations(cfg)
    db = synthetic.make_database(5, cfg, 64 * 30)
    eng = RefinementEngine(cfg, gpu, PatchDatabase(db['emb'], db['meta'], db['volumes'], gpu))
    sds = {}
    for name, m in eng.modules().items():
        sds[name] = helpers.seeded_sd({k: tuple(v.shape) for k, v in m.state_dict().items()}, 77 + len(name))
    eng.load_state_dicts(sds)
    raws = np.stack([synthetic.make_chunk(4000 + b, cfg)['input_raw'] for b in range(B)])
    x = torch.from_numpy(raws).to(gpu)
    rows = B * cfg['attn_num_patch'] ** 3
    noise = None
    if cfg['attn_retrieval_mode']:
        # large noise: the Gumbel-hard arg-max cannot flip on the ~1e-6 feature differences between the two MLP forms
        noise = (-torch.empty(rows, cfg['K']).exponential_(generator=torch.Generator().manual_seed(1)).log() * 4.0).to(gpu)
    fast = eng.refine(x, gumbel_noise=noise)
    saved = (ops.USE_CONV_UP, ops.USE_FUSED_POOL, ops.USE_FUSED_ATTN_MLP)
    ops.USE_CONV_UP, ops.USE_FUSED_POOL, ops.USE_FUSED_ATTN_MLP = False, False, False
    try:
        plain = eng.refine(x, gumbel_noise=noise)
    finally:
        ops.USE_CONV_UP, ops.USE_FUSED_POOL, ops.USE_FUSED_ATTN_MLP = saved
    assert torch.isfinite(fast).all()
    diff = (fast.cpu().double() - plain.cpu().double()).abs()
    err, tol = diff.max().item(), 0.2 * df_tolerance(trunc_t)
    over = int((diff > tol).sum().item())
    print(f'\n{cfg_name} B={B}: fast vs plain routes df max abs diff {err:.2e} (trunc {trunc_t}), voxels over {tol:.1e}: {over}')
    if cfg['attn_retrieval_mode']:
        # Gumbel-hard attention: the two routes' features differ by ~1e-6, and among the B * 4096 rows one near-tie of the (noise-dominated)
        # scores may resolve the other way; a flipped 2^3 patch reaches at most 8^3 output voxels.  Anything systematic moves far more.
        assert over <= 2 * 512, f'{over} voxels differ by more than {tol:.1e} (max {err:.2e})'
    else:
        assert err <= tol


@pytest.mark.parametrize('cfg_name,B', [('C1', 8), ('C5', 4)])
def test_engine_two_stream_schedule_is_bit_equal_to_serial(gpu, cfg_name, B):
    """The backbone runs on a second stream beside the retrieval path; both schedules must give the same bits (see tests/test_two_stream_gpu.py for
    the full matrix and DESIGN 4.7 for what once made them differ)."""
    from rfuse.database import PatchDatabase
    from rfuse.engine import RefinementEngine
    cfg = rf_configs.get_config(cfg_name)
    db = synthetic.make_database(6, cfg, 64 * 30)
    eng = RefinementEngine(cfg, gpu, PatchDatabase(db['emb'], db['meta'], db['volumes'], gpu))
    sds = {}
    for name, m in eng.modules().items():
        sds[name] = helpers.seeded_sd({k: tuple(v.shape) for k, v in m.state_dict().items()}, 91 + len(name))
    eng.load_state_dicts(sds)
    raws = np.stack([synthetic.make_chunk(5000 + b, cfg)['input_raw'] for b in range(B)])
    x = torch.from_numpy(raws).to(gpu)
    noise = None
    if cfg['attn_retrieval_mode']:
        rows = B * cfg['attn_num_patch'] ** 3
        noise = (-torch.empty(rows, cfg['K']).exponential_(generator=torch.Generator().manual_seed(2)).log()).to(gpu)
    eng.serial = True
    ref = eng.refine(x, gumbel_noise=noise).clone()
    eng.serial = False
    for _ in range(3):
        assert torch.equal(eng.refine(x, gumbel_noise=noise), ref)


@pytest.mark.parametrize('name', ['feat_C1', 'feat_C5'])
def test_get_features_matches_reference_golden(gpu, name):
    """A8: PatchedAttentionBlock.get_features as forward_full calls it (reference trainer/train_refinement.py:113-119,
    model/attention.py:132-139, 72-82): theta on the backbone features, phi on the TARGET's retrieval-backbone features,
    per-patch any() of the occupancy grid."""
    from model.attention import Unfold3D, Fold3D
    ffix = helpers.load_fixture(name)
    nfix = helpers.load_fixture(name.replace('feat_', 'net_'))
    cfg0 = rf_configs.get_config(str(nfix['cfg_name']))
    mods = build_modules(cfg0)
    shapes = {k: {n: tuple(v.shape) for n, v in m.state_dict().items()} for k, m in mods.items()}
    cfg, x_in, retr, sds = helpers.fixture_problem(nfix, shapes)
    for k, m in mods.items():
        m.load_state_dict(sds[k])
        m.to(gpu).eval()
    seed = int(ffix['seed'])
    tgt = np.stack([synthetic.normalise_target(cfg, synthetic.make_chunk(seed * 100 + 90, cfg)['target_raw'])[None]])
    assert helpers.sha(tgt) == str(ffix['target_sha'])
    occ = torch.from_numpy(np.unpackbits(ffix['occupancy'])[:32 ** 3].astype(bool).reshape(1, 1, 32, 32, 32)).to(gpu)
    with torch.no_grad():
        x_back = mods['unet_backbone'](torch.from_numpy(x_in).to(gpu))
        x_target = Fold3D(4, 8, cfg['nf'])(mods['retrieval_backbone'](Unfold3D(16, 1)(torch.from_numpy(tgt).to(gpu))))
        xf, pf, of = mods['patched_attention_block'].get_features(x_back, x_target, occ)
    assert maxerr(x_target[..., ::4, ::4, ::4].cpu(), ffix['x_target_sub']) <= 1e-4
    e_x, e_p = maxerr(xf.cpu(), ffix['x_feat']), maxerr(pf.cpu(), ffix['p_feat'])
    print(f'\n{name}: theta feats {e_x:.2e}  phi feats {e_p:.2e} (unit vectors)')
    assert tuple(xf.shape) == tuple(ffix['x_feat'].shape) and e_x <= 2e-5 and e_p <= 2e-5
    assert of.dtype == torch.bool
    np.testing.assert_array_equal(of.cpu().numpy(), np.unpackbits(ffix['occ_flat'])[:of.numel()].astype(bool))


def test_engine_patch_mask_matches_reference_compose(gpu):
    """A15: patches the query-side occupancy filter drops (reference dataset/patched_scene_dataset.py:28-32) are never looked
    up and keep the truncation value in all K retrieved volumes (util/retrieval.py:148,151).  Golden: the reference's own
    create_retrieval_from_mapping driven with a patch_from_scene_lookup that lacks those patches."""
    from rfuse import ops
    fix = helpers.load_fixture('retrieval_map_compose')
    cfg = rf_configs.get_config('C1')
    _, trunc_t = rf_configs.truncations(cfg)
    K = cfg['K']
    db = synthetic.make_database(int(fix['seed']), cfg, int(fix['n_patches']))
    keep = fix['patch_keep']
    q = torch.from_numpy(fix['queries']).to(gpu)
    from rfuse.database import PatchDatabase
    pdb = PatchDatabase(db['emb'], db['meta'], db['volumes'], gpu)
    meta, dist, idx = pdb.retrieve(q, K, None, torch.from_numpy(keep).to(gpu))
    m = meta.cpu().numpy()
    np.testing.assert_array_equal(m[keep], fix['map_val'][keep][..., :7].astype(np.int32))
    assert (m[~keep, :, 0] == -1).all() and (idx.cpu().numpy()[~keep] == -1).all() and torch.isinf(dist[~torch.from_numpy(keep).to(gpu)]).all()
    # composed volumes: un-normalised (mean 0, std 1) so the reference's raw output compares bit for bit
    meta[5, 1] = torch.tensor(fix['map_val_sentinel'][5, 1, :7].astype(np.int32))        # the fixture's sentinel hit
    composed = ops.gather_patches(pdb.volumes, meta, 1, K, trunc_t, 1.0, 0.0, 1.0, layout=0).cpu().numpy()[0]
    assert helpers.sha(composed) == str(fix['compose_masked_sha'])
    np.testing.assert_array_equal(composed[:, ::4, ::4, ::4], fix['compose_masked_sub'])


def test_half_voxel_store_gathers_the_same_bits(gpu):
    """PatchDatabase(half_store=True): the voxel store in the reference's own precision (float16 scenes, dataset/scene.py:61,71) -- half the bytes, the same
    gathered patches and the same refined chunks bit for bit; a store float16 cannot hold exactly is refused."""
    from rfuse import ops
    from rfuse.database import PatchDatabase
    from rfuse.engine import RefinementEngine
    cfg = rf_configs.get_config('C3')
    _, trunc_t = rf_configs.truncations(cfg)
    d, K = cfg['dataset_train'], cfg['K']
    db = synthetic.make_database(9, cfg, 64 * 12)
    full = PatchDatabase(db['emb'], db['meta'], db['volumes'], gpu, half_store=False)
    half = PatchDatabase(db['emb'], db['meta'], db['volumes'], gpu, half_store=True)
    auto = PatchDatabase(db['emb'], db['meta'], db['volumes'], gpu)                      # default: float16 when every voxel survives the round trip (it does here)
    assert full.volumes.dtype == torch.float32 and not full.half_store and auto.half_store and torch.equal(auto.volumes, half.volumes)
    assert half.volumes.dtype == torch.float16 and half.volumes.numel() * 2 == full.volumes.numel() * full.volumes.element_size() // 2
    q = torch.nn.functional.normalize(torch.randn(128, 64, generator=torch.Generator().manual_seed(4)), dim=1).to(gpu)
    meta, _, _ = full.retrieve(q, K)
    meta[3, 1, 0] = -1                                                # a "no neighbour" entry: truncation fill from either store
    for layout in (0, 1):
        a = ops.gather_patches(full.volumes, meta, 2, K, trunc_t, 1.0, d['target_mean'], d['target_std'], layout=layout)
        b = ops.gather_patches(half.volumes, meta, 2, K, trunc_t, 1.0, d['target_mean'], d['target_std'], layout=layout)
        assert torch.equal(a, b)
    raws = torch.from_numpy(np.stack([synthetic.make_chunk(440 + b, cfg)['input_raw'] for b in range(2)])).to(gpu)
    outs = []
    for pdb in (full, half):
        eng = RefinementEngine(cfg, gpu, pdb)
        eng.load_state_dicts({n: helpers.seeded_sd({k: tuple(v.shape) for k, v in m.state_dict().items()}, 210 + i) for i, (n, m) in enumerate(eng.modules().items())})
        outs.append(eng.refine(raws).clone())
    assert torch.equal(outs[0], outs[1])
    lossy = db['volumes'].copy()
    lossy[0, 0, 0, 0] = np.float32(0.1)                               # 0.1 is not a float16
    with pytest.raises(ValueError, match='float16 cannot represent'):
        PatchDatabase(db['emb'], db['meta'], lossy, gpu, half_store=True)
    kept = PatchDatabase(db['emb'], db['meta'], lossy, gpu)                               # ... and the default keeps such a store as float32
    assert kept.volumes.dtype == torch.float32 and not kept.half_store
    nan = db['volumes'].copy()
    nan[1, 2, 3, 4] = np.nan                                                              # a NaN voxel is float16-representable: not a reason to refuse
    assert PatchDatabase(db['emb'], db['meta'], nan, gpu, half_store=True).volumes.dtype == torch.float16


def test_engine_patch_mask_end_to_end(gpu):
    """refine(patch_mask=...) == the oracle composing with the same dropped patches, then the networks"""
    from rfuse.database import PatchDatabase
    from rfuse.engine import RefinementEngine
    cfg = rf_configs.get_config('C3')
    trunc_i, trunc_t = rf_configs.truncations(cfg)
    K, B = cfg['K'], 2
    db = synthetic.make_database(22, cfg, 64 * 20)
    eng = RefinementEngine(cfg, gpu, PatchDatabase(db['emb'], db['meta'], db['volumes'], gpu))
    sds = {n: helpers.seeded_sd({k: tuple(v.shape) for k, v in m.state_dict().items()}, 500 + len(n)) for n, m in eng.modules().items()}
    eng.load_state_dicts(sds)
    raws = np.stack([synthetic.make_chunk(800 + b, cfg)['input_raw'] for b in range(B)])
    mask = np.random.default_rng(3).random((B, 64)) > 0.5
    df = eng.refine(torch.from_numpy(raws).to(gpu), patch_mask=torch.from_numpy(mask).to(gpu)).cpu().numpy()
    with torch.no_grad():
        q = torch.cat([refpath.embed_queries(refpath.extract_query_windows(r, cfg, trunc_i), sds['fenc_input'], cfg) for r in raws]).numpy()
        idx, dist = refpath.knn_exact(q, db['emb'], 2 * K)
        mapping = refpath.demote_same_scene(refpath.mapping_rows(idx, dist, db['meta']), np.full(B * 64, -1), K)
        d = cfg['dataset_train']
        retr = np.stack([refpath.compose_retrieval(mapping[b * 64:(b + 1) * 64], db['volumes'], K, trunc_t, patch_keep=mask[b]) for b in range(B)])
        retr = ((retr - np.float32(d['target_mean'])) / np.float32(d['target_std'])).astype(np.float32)
        x_in = np.stack([synthetic.normalise_input(cfg, r)[None] for r in raws])
        ref = refpath.forward_full(sds, cfg, torch.from_numpy(x_in), torch.from_numpy(retr), trunc_t).numpy()
        full = eng.refine(torch.from_numpy(raws).to(gpu)).cpu().numpy()
    assert maxerr(df, ref) <= df_tolerance(trunc_t)
    assert maxerr(df, full) > 1e-3, 'the mask must change the result'


def test_engine_on_a_non_current_device():
    """RefinementEngine(cfg, 'cuda:1', db) with current device 0: every launch, stream and workspace must follow the tensors'
    device (_stream() used the current device).  Needs two GPUs; skipped on the 1-GPU test box."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs')
    from rfuse.database import PatchDatabase
    from rfuse.engine import RefinementEngine
    cfg = rf_configs.get_config('C3')
    db = synthetic.make_database(3, cfg, 640)
    raws = torch.from_numpy(np.stack([synthetic.make_chunk(123 + b, cfg)['input_raw'] for b in range(2)]))
    outs = []
    torch.cuda.set_device(0)
    for dev in ('cuda:0', 'cuda:1'):
        eng = RefinementEngine(cfg, dev, PatchDatabase(db['emb'], db['meta'], db['volumes'], dev))
        eng.load_state_dicts({n: helpers.seeded_sd({k: tuple(v.shape) for k, v in m.state_dict().items()}, 70 + i)
                              for i, (n, m) in enumerate(eng.modules().items())})
        assert torch.cuda.current_device() == 0
        outs.append(eng.refine(raws.to(dev)).cpu())
        torch.cuda.synchronize(dev)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('cfg_name,B,chunks', [('C3', 32, (0, 13, 31)), ('C2', 32, (0, 13, 31)), ('C4', 16, (0, 15)), ('C5', 16, (0, 15))])
def test_engine_at_the_bench_batch_size_matches_oracle(gpu, cfg_name, B, chunks):
    """bench.py runs B = 32 chunks per step, where the big-tile / position-major / parity-split kernels are selected
    (dispatch depends on the sample count); the kernel tests cover them one by one, this one checks the whole online path at
    that batch size against the oracle on a few chunks of the batch (GroupNorm is per sample: a chunk's result does not
    depend on its batch mates).  C2 / C5 are Gumbel-hard: explicit noise, scaled so the arg-max cannot flip on fp32 differences.
    C4 (Matterport3D: K = 8, trunc 11.25, softmax sharpness 1024) and C5 (point-cloud encoder, nf = 12 U-Net on a 128^3 grid) run at the
    batch size their bench lines use (16).  Bar: north_star's 1e-4 against the float64 oracle; where the oracle's OWN fp32 evaluation is
    further than that from its float64 evaluation on these very chunks (C4), the HIP path must be no further from the truth than 1.25 x
    that distance."""
    from rfuse.database import PatchDatabase
    from rfuse.engine import RefinementEngine
    cfg = rf_configs.get_config(cfg_name)
    trunc_i, trunc_t = rf_configs.truncations(cfg)
    K = cfg['K']
    db = synthetic.make_database(31, cfg, 64 * 50)
    eng = RefinementEngine(cfg, gpu, PatchDatabase(db['emb'], db['meta'], db['volumes'], gpu))
    sds = {n: helpers.seeded_sd({k: tuple(v.shape) for k, v in m.state_dict().items()}, 1300 + len(n)) for n, m in eng.modules().items()}
    eng.load_state_dicts(sds)
    raws = np.stack([synthetic.make_chunk(5000 + b, cfg)['input_raw'] for b in range(B)])
    rows = cfg['attn_num_patch'] ** 3
    noise = None
    if cfg['attn_retrieval_mode']:
        noise = -torch.empty(B * rows, K).exponential_(generator=torch.Generator().manual_seed(5)).log() * 4.0
    df = eng.refine(torch.from_numpy(raws).to(gpu), gumbel_noise=noise.to(gpu) if noise is not None else None).cpu().numpy()
    d = cfg['dataset_train']
    sds64 = {m: {k: v.double() for k, v in sd.items()} for m, sd in sds.items()}
    worst = worst32 = oracle32 = 0.0
    for b in chunks:
        with torch.no_grad():
            q = refpath.embed_queries(refpath.extract_query_windows(raws[b], cfg, trunc_i), sds['fenc_input'], cfg).numpy()
            idx, dist = refpath.knn_exact(q, db['emb'], 2 * K)
            mapping = refpath.demote_same_scene(refpath.mapping_rows(idx, dist, db['meta']), np.full(64, -1), K)
            retr = refpath.compose_retrieval(mapping, db['volumes'], K, trunc_t)[None]
            retr = ((retr - np.float32(d['target_mean'])) / np.float32(d['target_std'])).astype(np.float32)
            x_in = synthetic.normalise_input(cfg, raws[b])[None, None]
            nb = noise[b * rows:(b + 1) * rows] if noise is not None else None
            ref32 = refpath.forward_full(sds, cfg, torch.from_numpy(x_in), torch.from_numpy(retr), trunc_t, nb).numpy()
            # the oracle in float64 = the truth; its own fp32 evaluation (what the reference computes) is printed next to it
            ref64 = refpath.forward_full(sds64, cfg, torch.from_numpy(x_in).double(), torch.from_numpy(retr).double(), trunc_t,
                                         nb.double() if nb is not None else None).numpy()
        worst = max(worst, maxerr(df[b:b + 1], ref64))
        worst32 = max(worst32, maxerr(df[b:b + 1], ref32))
        oracle32 = max(oracle32, maxerr(ref32, ref64))
    print(f'\n{cfg_name} B={B}, chunks {chunks}: df max abs err vs float64 oracle {worst:.2e}, vs fp32 oracle {worst32:.2e} (fp32 oracle vs float64: {oracle32:.2e})')
    bar = max(DF_TOL, 1.25 * oracle32)
    assert worst <= bar and worst32 <= bar + oracle32
    if trunc_t > 1.0:                                                     # ... and north_star's 1e-4 on the tanh output itself (df error * 2 / trunc), see assert_df_parity
        print(f'   tanh output: max |pred_gpu - pred_oracle_fp32| = {worst32 * 2 / trunc_t:.2e}')
        assert worst32 * 2.0 / trunc_t <= DF_TOL


def test_gumbel_hard_path_with_unscaled_noise(gpu):
    """The Gumbel-hard attention (ShapeNetV2 configs, model/attention.py:100-103: hard arg-max of 25 * scores + g, g ~ Gumbel(0, 1)) END TO END with the
    noise at its real scale.  The other end-to-end comparisons multiply the noise by 4 so that no row's arg-max can flip on fp32 differences; here only
    the rows that really ARE near ties are taken out: the float64 evaluation of the same network gives every row's logits, rows whose two best differ
    by less than GAP get their winner's noise raised by 1 on BOTH sides (recorded, printed), every other row keeps its draw.  GAP = 2e-4: the scores
    are dot products of unit 32-vectors out of a 4-layer fp32 MLP (error ~ 2e-6, test_attn_mlp_rows_matches_float64) times the sharpness 25."""
    fix = helpers.load_fixture('net_C1')
    cfg0 = rf_configs.get_config(str(fix['cfg_name']))
    assert cfg0['attn_retrieval_mode']
    mods = build_modules(cfg0)
    shapes = {k: {n: tuple(v.shape) for n, v in m.state_dict().items()} for k, m in mods.items()}
    cfg, x_in, retr, sds = helpers.fixture_problem(fix, shapes)
    for k, m in mods.items():
        m.load_state_dict(sds[k])
        m.to(gpu).eval()
    K, B = cfg['K'], x_in.shape[0]
    trunc_t = float(fix['target_trunc'])
    GAP = 2e-4
    noise = -torch.empty(B * cfg['attn_num_patch'] ** 3, K).exponential_(generator=torch.Generator().manual_seed(77)).log()      # unscaled
    sds64 = {m: {k: v.double() for k, v in sd.items()} for m, sd in sds.items()}
    torch.set_num_threads(16)
    with torch.no_grad():
        xb = refpath.unet_backbone(torch.from_numpy(x_in).double(), sds64['unet_backbone'], cfg)
        r64 = torch.from_numpy(retr).double()[:, :K].reshape(B * K, 1, 64, 64, 64)
        xr = refpath.fold3d(refpath.retrieval_backbone(refpath.unfold3d(r64, 16), sds64['retrieval_backbone'], cfg), 4, 8, cfg['nf'])
        det = {}
        refpath.patched_attention_block(xb, xr, sds64['patched_attention_block'], cfg, noise.double(), det)
        logits = det['scores'] * 25 + noise.double()
        top2 = logits.topk(2, dim=1)
        near = (top2.values[:, 0] - top2.values[:, 1]) < GAP
        noise[near, top2.indices[near, 0]] += 1.0                       # the near ties decided the same way for everyone
        ref64 = refpath.forward_full(sds64, cfg, torch.from_numpy(x_in).double(), torch.from_numpy(retr).double(), trunc_t, noise.double()).numpy()
        from model.attention import Unfold3D
        x_back = mods['unet_backbone'](torch.from_numpy(x_in).to(gpu))
        feats = mods['retrieval_backbone'](Unfold3D(16, 1)(torch.from_numpy(retr).to(gpu)[:, :K].reshape(B * K, 1, 64, 64, 64)))
        x_attn = mods['patched_attention_block'].forward_patch_major(x_back, feats, 8, noise.to(gpu))
        df = mods['decoder'].forward_df(x_attn, trunc_t).cpu().numpy()
    gaps = (top2.values[:, 0] - top2.values[:, 1])[~near]
    err = maxerr(df, ref64)
    print(f'\nunscaled Gumbel noise: {int(near.sum())} of {near.numel()} rows within {GAP} of a tie (decided explicitly); smallest remaining gap {float(gaps.min()):.2e}; '
          f'df max abs err vs float64 {err:.2e}')
    assert int(near.sum()) <= 0.002 * near.numel()
    assert err <= DF_TOL


def test_scene_driver_matches_chunkwise_refinement(gpu):
    """N3: rfuse.scene.refine_scene (scene -> chunk grid -> batched RefinementEngine.refine -> float16 -> recomposition; the reference's vis_infer
    loop, trainer/train_refinement.py:158-169) against refining every chunk on its own and pasting it by hand."""
    from rfuse import scene
    from rfuse.database import PatchDatabase
    from rfuse.engine import RefinementEngine
    cfg = rf_configs.get_config('C3')
    trunc_i, trunc_t = rf_configs.truncations(cfg)
    db = synthetic.make_database(8, cfg, 64 * 20)
    eng = RefinementEngine(cfg, gpu, PatchDatabase(db['emb'], db['meta'], db['volumes'], gpu))
    sds = {n: helpers.seeded_sd({k: tuple(v.shape) for k, v in m.state_dict().items()}, 500 + len(n)) for n, m in eng.modules().items()}
    eng.load_state_dicts(sds)
    low = np.concatenate([np.concatenate([synthetic.make_chunk(900 + 2 * i + j, cfg)['input_raw'] for j in range(2)], axis=1) for i in range(2)], axis=0)
    low = np.concatenate([low, synthetic.make_chunk(950, cfg)['input_raw'][:, :, :4].repeat(2, axis=0).repeat(2, axis=1)], axis=2)     # [16, 16, 12]: ragged in z
    names, chunks = scene.split_scene(low, 8, 'sceneQ', pad_value=trunc_i)
    assert chunks.shape[0] == 8
    out = scene.refine_scene(eng, names, chunks, batch=3)
    assert list(out) == ['sceneQ__room0'] and out['sceneQ__room0'].shape == (128, 128, 128)
    ref = np.full((128, 128, 128), trunc_t, dtype=np.float64)
    for name, ch in zip(names, chunks):
        x, y, z = [int(t) for t in name.split('__')[-1].split('_')]
        ref[x:x + 64, y:y + 64, z:z + 64] = eng.refine(torch.from_numpy(ch[None]).to(gpu))[0, 0].half().cpu().numpy()
    assert np.abs(out['sceneQ__room0'] - ref).max() <= 2e-4                    # batch size changes the tile dispatch, not the arithmetic contract; float16 steps are 1.2e-4 at trunc
    # the canvas assembled on the device (default) and the host-side pasting of float16 chunks (combine_predictions, what the goldens pin) give the same array
    host = scene.refine_scene(eng, names, chunks, batch=3, assemble_on_device=False)
    assert out['sceneQ__room0'].dtype == np.float64 and np.array_equal(out['sceneQ__room0'], host['sceneQ__room0'])
    # several superscenes in one call, a chunk pasted twice (the later one wins)
    names2 = [n.replace('sceneQ', 'sceneR') for n in names[:3]] + names + [names[0]]
    chunks2 = np.concatenate([chunks[:3], chunks, chunks[5:6]])
    both = scene.refine_scene(eng, names2, chunks2, batch=4)
    ref2 = scene.refine_scene(eng, names2, chunks2, batch=4, assemble_on_device=False)
    assert list(both) == ['sceneR__room0', 'sceneQ__room0'] and all(np.array_equal(both[k], ref2[k]) for k in both)
    # many scenes through ONE pipelined stream (refine_scenes: a scene is handed out while the next one is being refined): each equals its own refine_scene call
    # (scenes without chunks -- first, in the middle, twice in a row, last -- yield {} at their position: ADVICE r4, the results must stay zip-able with the scenes)
    none = ([], np.zeros((0, 8, 8, 8), np.float32))
    seq = [none, (names, chunks), (names2, chunks2), none, none, (names[:2], chunks[:2]), (names, chunks), none]
    streamed = list(scene.refine_scenes(eng, iter(seq), batch=3))
    assert len(streamed) == len(seq)
    for (nm, ch), got in zip(seq, streamed):
        if not len(nm):
            assert got == {}
            continue
        want = scene.refine_scene(eng, nm, ch, batch=3)
        assert list(got) == list(want) and all(np.array_equal(got[k], want[k]) for k in want)
    # a negative origin would be an offset in front of the canvas buffer on the device: refused when the scene is laid out
    with pytest.raises(ValueError, match='negative origin'):
        scene.refine_scene(eng, ['sceneN__room0__-64_0_0', 'sceneN__room0__0_0_0'], chunks[:2], batch=2)


@pytest.mark.parametrize('cfg_name,B', [('C3', 8), ('C4', 4)])
def test_refine_stream_is_bit_equal_to_refine(gpu, cfg_name, B):
    """RefinementEngine.refine_stream (front end of batch i + 1 on a helper stream beside the back end of batch i) runs the same kernels on the same
    data as refine(): every batch of a stream of different batches (with same-scene demotion and a query-side patch mask on some) must come out
    bit for bit as from refine() alone."""
    from rfuse.database import PatchDatabase
    from rfuse.engine import RefinementEngine
    cfg = rf_configs.get_config(cfg_name)
    db = synthetic.make_database(12, cfg, 64 * 30)
    eng = RefinementEngine(cfg, gpu, PatchDatabase(db['emb'], db['meta'], db['volumes'], gpu))
    sds = {n: helpers.seeded_sd({k: tuple(v.shape) for k, v in m.state_dict().items()}, 300 + len(n)) for n, m in eng.modules().items()}
    eng.load_state_dicts(sds)
    batches, scenes, masks = [], [], []
    for i in range(5):
        batches.append(torch.from_numpy(np.stack([synthetic.make_chunk(6000 + 10 * i + b, cfg)['input_raw'] for b in range(B)])).to(gpu))
        scenes.append(torch.full((B * 64,), i % 3 - 1, dtype=torch.int32, device=gpu) if i % 2 else None)
        m = torch.ones(B, 64, dtype=torch.bool, device=gpu)
        m[:, ::5] = False
        masks.append(m if i in (1, 4) else None)
    want = [eng.refine(x, query_scene=s, patch_mask=m).clone() for x, s, m in zip(batches, scenes, masks)]
    got = list(eng.refine_stream(batches, scenes, masks))
    assert len(got) == 5
    for i, (g, w) in enumerate(zip(got, want)):
        assert torch.equal(g, w), 'batch %d differs' % i


def test_clumped_database_is_searched_by_the_data_independent_scan(gpu):
    """A shard with clumps of near-identical rows (thousands of copies of an empty-space patch; a collapsed encoder) would make every row of a clump a survivor of the
    matrix-core filter for every query near it: PatchDatabase probes the shard once and takes the VALU scan for it -- the same lists, a cost that does not depend on
    the data."""
    from rfuse import ops
    from rfuse.database import PatchDatabase
    g = torch.Generator().manual_seed(5)
    n = 64 * 300
    emb = torch.nn.functional.normalize(torch.randn(n + 1, 64, generator=g), dim=1)
    meta = torch.zeros(n + 1, 7, dtype=torch.int32)
    vols = torch.zeros(300, 64, 64, 64, dtype=torch.float16)
    plain = PatchDatabase(emb, meta, vols, gpu)
    assert plain.scan_algo == 0
    clumped = emb.clone()
    clumped[torch.randperm(n, generator=g)[: n // 3]] = torch.nn.functional.normalize(emb[7] + 1e-4 * torch.randn(n // 3, 64, generator=g), dim=1)
    db = PatchDatabase(clumped, meta, vols, gpu)
    assert db.scan_algo == ops.TOPK_VALU_SCAN
    q = torch.cat([clumped[7:8], torch.nn.functional.normalize(torch.randn(127, 64, generator=g), dim=1)]).to(gpu)
    d, i = db.search(q, 8)
    d3, i3 = ops.l2_topk(q, db.emb_packed, n + 1, 0, 8, ops.TOPK_MFMA16_SCAN)
    assert torch.equal(i, i3) and torch.equal(d, d3)
