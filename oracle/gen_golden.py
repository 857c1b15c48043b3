"""Golden-vector generator -- runs ONLY in the build container, where /root/reference exists.

Imports the reference's own ``model`` package (pure PyTorch) and, under ``sys.modules`` stubs for the
dependencies this image lacks (pyflann, trimesh, marching_cubes, pyrender, torchmetrics, the un-vendored Chamfer
submodule), its ``util.retrieval`` module; runs them on seeded synthetic inputs with seeded weights and writes
small ``.npz`` fixtures to tests/golden/.  Fixtures hold DATA only (inputs' digests, expected outputs); no
reference source travels.

    python oracle/gen_golden.py            # regenerate everything
"""
import hashlib
import json
import os
import sys
import tempfile
import types
from pathlib import Path
from unittest import mock

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
REF = Path('/root/reference')
OUT = REPO / 'tests' / 'golden'

sys.path.insert(0, str(REPO / 'retrieval-fuse_amd'))
from rfuse import configs as rf_configs            # noqa: E402  (numpy-only product helpers: configs + generators)
from rfuse import synthetic                        # noqa: E402


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def sub(t, stride):
    """spatial subsample of an NCDHW tensor"""
    return t[..., ::stride, ::stride, ::stride].contiguous().numpy()


def stats(t):
    t64 = t.double()
    return np.array([t64.sum().item(), t64.abs().sum().item(), (t64 * t64).sum().item()], dtype=np.float64)


def save_fixture(name, **arrays):
    """np.savez_compressed, but an existing fixture whose arrays are all identical is left untouched (zip members carry
    timestamps, so re-writing would churn the committed bytes for nothing)."""
    path = OUT / (name + '.npz')
    if path.exists():
        old = np.load(path, allow_pickle=False)
        same = sorted(old.files) == sorted(arrays)
        if same:
            for k in old.files:
                a, b = old[k], np.asarray(arrays[k])
                if a.shape != b.shape or a.dtype != b.dtype or not np.array_equal(a, b, equal_nan=(a.dtype.kind == 'f')):
                    same = False
                    break
        if same:
            print('  (unchanged)', path.name)
            return
    np.savez_compressed(path, **arrays)


def import_reference_model():
    # our product package also has a top-level ``model``; make sure the reference's wins in this process
    for k in [k for k in sys.modules if k == 'model' or k.startswith('model.')]:
        del sys.modules[k]
    sys.path.insert(0, str(REF))
    import model as ref_model
    assert str(REF) in ref_model.__file__, ref_model.__file__
    return ref_model


def load_seeded(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synthetic.seeded_state_dict(shapes, seed)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    module.eval()
    return sd


def chunk_inputs(cfg, seed, batch, stress=False):
    """x_in [B,1,S,S,S] normalised; retr [B,K,64,64,64] normalised (K other synthetic chunks per sample)."""
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


def gen_network_fixture(ref_model, name, cfg_name, seed, batch, stress=False, with_features=False):
    cfg = rf_configs.get_config(cfg_name)
    trunc_i, trunc_t = rf_configs.truncations(cfg)
    with mock.patch('builtins.print'):
        unet = ref_model.get_unet_backbone(cfg)
        dec = ref_model.get_decoder(cfg)
        rb = ref_model.get_retrieval_backbone(cfg)
        pab = ref_model.get_attention_block(cfg)
    w_digest = sha(*[v for m, s in ((unet, 11), (dec, 12), (rb, 13), (pab, 14)) for v in load_seeded(m, seed * 1000 + s).values()])
    from model.attention import Unfold3D, Fold3D
    unfold_shape, fold_features = Unfold3D(16, 1), Fold3D(4, 8, rb.nf)

    x_in, retr = chunk_inputs(cfg, seed, batch, stress)
    K = cfg['K']
    out = {'cfg_name': cfg_name, 'seed': seed, 'batch': batch, 'stress': int(stress), 'weights_sha': w_digest,
           'inputs_sha': sha(x_in, retr), 'target_trunc': np.float32(trunc_t)}
    with torch.no_grad():
        xt, rt = torch.from_numpy(x_in), torch.from_numpy(retr)
        x_back = unet(xt)                                                # trainer/train_refinement.py:109
        retrievals = rt[:, :K].reshape(batch * K, 1, 64, 64, 64)         # :255-257
        x_retr = fold_features(rb(unfold_shape(retrievals)))             # :112
        noise_seed = seed * 7 + 1
        min_margin = np.inf
        if cfg['attn_retrieval_mode']:
            # the Gumbel draw is the first RNG use inside the attention forward: reproduce it with the same seed
            while True:
                torch.manual_seed(noise_seed)
                nrows = batch * cfg['attn_num_patch'] ** 3
                noise = -torch.empty(nrows, K, memory_format=torch.legacy_contiguous_format).exponential_().log()
                torch.manual_seed(noise_seed)
                x_attn = pab(x_back, x_retr)                             # :115
                # margin of the hard arg-max, from an independent evaluation of the logits
                sys.path.insert(0, str(REPO))
                from oracle import refpath
                det = {}
                sdp = {k: v for k, v in pab.state_dict().items()}
                x_chk = refpath.patched_attention_block(x_back, x_retr, sdp, cfg, noise, det)
                top2 = torch.topk(det['scores'] * 25 + noise, 2, dim=1).values
                min_margin = float((top2[:, 0] - top2[:, 1]).min())
                assert torch.equal(x_chk, x_attn), 'captured Gumbel noise does not reproduce the reference forward'
                if min_margin > 1e-4:
                    break
                noise_seed += 1
            out['gumbel_noise'] = noise.numpy()
            out['noise_seed'] = noise_seed
        else:
            x_attn = pab(x_back, x_retr)
        pred = dec(x_attn)                                               # :116
        df = (pred + 1) * trunc_t / 2                                    # :242-243
    out.update(
        min_margin=min_margin,
        x_back_sub=sub(x_back, 2), x_back_stats=stats(x_back),
        x_retr_sub=sub(x_retr, 4), x_retr_stats=stats(x_retr),
        x_attn_sub=sub(x_attn, 2), x_attn_stats=stats(x_attn),
        pred_stats=stats(pred), df=df.numpy().astype(np.float32),
    )
    save_fixture(name, **out)
    print(name, 'df', df.shape, 'margin', min_margin, 'x_back absmax', float(x_back.abs().max()), 'pred range', float(pred.min()), float(pred.max()))
    gen_truth_fixture(name, cfg, (unet, dec, rb, pab), xt, rt, trunc_t, out.get('gumbel_noise'), df, x_back, x_retr, x_attn)
    if with_features:
        gen_features_fixture(name.replace('net_', 'feat_'), cfg, seed, (unet, dec, rb, pab), xt, x_back, trunc_t)


def gen_truth_fixture(name, cfg, mods, xt, rt, trunc_t, noise, df_ref, x_back_ref, x_retr_ref, x_attn_ref):
    """float64 evaluation of the SAME network on the SAME inputs = the 'truth' both fp32 implementations (the reference's
    ATen-CPU fp32 and the HIP path) are measured against.  Computed with the oracle in float64; for the deterministic
    (softmax) configs the reference's own modules are also run in .double() and must agree with it to 1e-9."""
    sys.path.insert(0, str(REPO))
    from oracle import refpath
    unet, dec, rb, pab = mods
    sds64 = {n: {k: v.double() for k, v in m.state_dict().items()} for n, m in
             (('unet_backbone', unet), ('decoder', dec), ('retrieval_backbone', rb), ('patched_attention_block', pab))}
    st = {}
    with torch.no_grad():
        df64 = refpath.forward_full(sds64, cfg, xt.double(), rt.double(), float(trunc_t),
                                    torch.from_numpy(noise).double() if noise is not None else None, st)
        ref_double_gap = -1.0
        if not cfg['attn_retrieval_mode']:
            import copy
            from model.attention import Unfold3D, Fold3D
            u64, d64, r64, p64 = (copy.deepcopy(m).double() for m in mods)
            b, K = xt.shape[0], cfg['K']
            xb = u64(xt.double())
            xr = Fold3D(4, 8, rb.nf)(r64(Unfold3D(16, 1)(rt.double()[:, :K].reshape(b * K, 1, 64, 64, 64))))
            dfr = (d64(p64(xb, xr)) + 1) * float(trunc_t) / 2
            ref_double_gap = float((dfr - df64).abs().max())
            assert ref_double_gap <= 1e-9, ref_double_gap
    err = (df_ref.double() - df64).abs()
    def resid(t64, t32):
        # the truth is stored as its float32 residual against the reference's fp32 result of the main fixture:
        # truth = float64(ref) + float64(residual); the residual is ~1e-6, so its own float32 rounding is ~1e-13
        return (t64 - t32.double()).float()
    out = dict(
        df_resid=resid(df64, df_ref).numpy(), ref_double_gap=ref_double_gap,
        x_back_resid_sub=sub(resid(st['x_back'], x_back_ref), 2), x_retr_resid_sub=sub(resid(st['x_retrieval'], x_retr_ref), 4),
        x_attn_resid_sub=sub(resid(st['x_attn'], x_attn_ref), 2),
        # how far the reference's own fp32 arithmetic sits from the truth (informational; tests recompute it from df / df_f64)
        ref_err_max=float(err.max()), ref_err_rms=float((err ** 2).mean().sqrt()),
        ref_frac_gt_1e4=float((err > 1e-4).double().mean()),
        ref_err_back=float((x_back_ref.double() - st['x_back']).abs().max()),
        ref_err_retr=float((x_retr_ref.double() - st['x_retrieval']).abs().max()),
        ref_err_attn=float((x_attn_ref.double() - st['x_attn']).abs().max()),
    )
    save_fixture(name.replace('net_', 'truth_'), **out)
    print('  truth', name, 'ref fp32 vs f64: df max %.3e rms %.3e frac>1e-4 %.5f | x_back %.2e x_retr %.2e x_attn %.2e | ref.double gap %.1e'
          % (out['ref_err_max'], out['ref_err_rms'], out['ref_frac_gt_1e4'], out['ref_err_back'], out['ref_err_retr'], out['ref_err_attn'],
             ref_double_gap))


def gen_features_fixture(name, cfg, seed, mods, xt, x_back, trunc_t):
    """A8: PatchedAttentionBlock.get_features exactly as forward_full calls it (trainer/train_refinement.py:113-119):
    theta on the backbone features, phi on the TARGET's retrieval-backbone features, occupancy from decoder(x_back)."""
    from model.attention import Unfold3D, Fold3D
    unet, dec, rb, pab = mods
    b = xt.shape[0]
    tgt = np.stack([synthetic.normalise_target(cfg, synthetic.make_chunk(seed * 100 + 90 + i, cfg)['target_raw'])[None] for i in range(b)])
    with torch.no_grad():
        x_target = Fold3D(4, 8, rb.nf)(rb(Unfold3D(16, 1)(torch.from_numpy(tgt))))
        df_back = (dec(x_back) + 1) * trunc_t / 2                                               # :118, :242-243
        voxel = cfg['dataset_train']['voxel_size_target']
        # occupancy_from_prediction, trainer/train_refinement.py:245-247; the reference thresholds with the float16-rounded voxel size
        vs = float(np.float16(voxel).astype(np.float32))
        occ = torch.nn.functional.max_pool3d((df_back <= vs * 0.75).float(), kernel_size=2, stride=2).bool()
        x_feat, p_feat, occ_flat = pab.get_features(x_back, x_target, occ)
    save_fixture(name, cfg_name=str(cfg_name_of(cfg)), seed=seed, target_sha=sha(tgt), occupancy=np.packbits(occ.numpy().reshape(-1)),
                 x_feat=x_feat.numpy(), p_feat=p_feat.numpy(), occ_flat=np.packbits(occ_flat.numpy()),
                 x_target_sub=sub(x_target, 4))
    print('  features', name, tuple(x_feat.shape), tuple(p_feat.shape), 'occupied patches', int(occ_flat.sum()))


def cfg_name_of(cfg):
    for k, v in rf_configs.CONFIGS.items():
        if v == cfg:
            return k
    raise KeyError


def gen_query_fixture(ref_model, cfg_name, seed):
    """A11: query windows -> fenc_input -> L2 normalise (util/retrieval.py:66)."""
    cfg = rf_configs.get_config(cfg_name)
    trunc_i, _ = rf_configs.truncations(cfg)
    fenc_input, _ = ref_model.get_retrieval_networks(cfg['retrieval_model'])
    load_seeded(fenc_input, seed * 1000 + 21)
    raw = synthetic.make_chunk(seed * 100, cfg)['input_raw']
    sys.path.insert(0, str(REPO))
    from oracle import refpath
    windows = refpath.extract_query_windows(raw, cfg, trunc_i)
    with torch.no_grad():
        z = fenc_input(torch.from_numpy(windows))
        lat = cfg['retrieval_model']['latent_dim']
        emb = torch.nn.functional.normalize(z.permute((0, 2, 3, 4, 1)).reshape((-1, lat)), dim=1)
    save_fixture('query_%s' % cfg_name, cfg_name=cfg_name, seed=seed, windows_sha=sha(windows),
                        windows_shape=np.array(windows.shape), emb=emb.numpy())
    print('query', cfg_name, windows.shape, emb.shape)


# ------------------------------------------------------------------ reference util/retrieval.py under stubs

class _ExactFLANN:
    """Brute-force stand-in for pyflann.FLANN: the interface the reference touches (load_index / nn_index)."""

    def load_index(self, path, pts):
        self.pts = np.asarray(pts, dtype=np.float64)

    def nn_index(self, q, n, checks=None):
        q = np.asarray(q, dtype=np.float64)
        d = ((q[:, None, :] - self.pts[None, :, :]) ** 2).sum(-1)
        order = np.argsort(d, axis=1, kind='stable')[:, :n]
        return order.astype(np.int32), np.take_along_axis(d, order, axis=1).astype(np.float32)


def import_reference_util_retrieval():
    stub_names = ['pyflann', 'trimesh', 'trimesh.sample', 'trimesh.voxel', 'trimesh.voxel.ops', 'marching_cubes', 'pyrender',
                  'torchmetrics', 'torchmetrics.metric', 'external', 'external.ChamferDistancePytorch',
                  'external.ChamferDistancePytorch.chamfer3D', 'PIL', 'PIL.Image']
    for n in stub_names:
        if n not in sys.modules:
            try:
                __import__(n)
            except Exception:
                sys.modules[n] = types.ModuleType(n)
    sys.modules['pyflann'].FLANN = _ExactFLANN
    sys.modules['pyflann'].__all__ = ['FLANN']
    if not hasattr(sys.modules['torchmetrics.metric'], 'Metric'):
        sys.modules['torchmetrics.metric'].Metric = type('Metric', (torch.nn.Module,), {})
    if not hasattr(sys.modules['external.ChamferDistancePytorch.chamfer3D'], 'dist_chamfer_3D'):
        sys.modules['external.ChamferDistancePytorch.chamfer3D'].dist_chamfer_3D = types.ModuleType('dist_chamfer_3D')
    tm = sys.modules['trimesh']
    for attr, val in (('sample', sys.modules['trimesh.sample']), ('voxel', sys.modules['trimesh.voxel'])):
        if not hasattr(tm, attr):
            setattr(tm, attr, val)
    for k in [k for k in sys.modules if k == 'util' or k.startswith('util.') or k == 'dataset' or k.startswith('dataset.')]:
        del sys.modules[k]
    sys.path.insert(0, str(REF))
    import util.retrieval as ref_ret
    assert str(REF) in ref_ret.__file__
    return ref_ret


class _FakeDataset:
    """The handful of attributes create_retrieval_from_mapping touches (util/retrieval.py:145-164)."""

    def __init__(self, volumes, scene_names, target_trunc, lookup):
        self.volumes, self.scene_names = volumes, scene_names
        self.target_trunc = np.float32(target_trunc)
        self.patch_from_scene_lookup = lookup
        self.no_overlap = True

    def get_scene_size(self, scene):
        return [64, 64, 64]

    def get_scene_target(self, scene):
        return self.volumes[self.scene_names.index(scene)]

    def unpad(self, *e):
        # patch_context_target = 8 (config/base/retrieval_superresolution.yaml:12), dataset/patched_scene_dataset.py:103-107
        if len(e) == 2:
            return [e[0], e[1] - 16]
        return self.unpad(e[0], e[1]) + self.unpad(e[2], e[3]) + self.unpad(e[4], e[5])


def gen_retrieval_fixture(seed=5):
    ref_ret = import_reference_util_retrieval()
    from dataset.scene import SceneHandler
    cfg = rf_configs.get_config('C1')
    _, trunc_t = rf_configs.truncations(cfg)
    K = cfg['K']
    n_patches = 64 * 12
    db = synthetic.make_database(seed, cfg, n_patches)
    scene_names = ['scene%03d' % i for i in range(db['n_scenes'])]
    rng = np.random.default_rng(seed + 1)
    # queries: scene 3 of the train split (present in the DB index -> demotion active), 64 patches;
    # make some of them close to DB rows of their own scene so demotion really reorders
    q_scene = 3
    queries = rng.standard_normal((64, 64)).astype(np.float32)
    own = np.where(db['meta'][:, 0] == q_scene)[0]
    for i in range(0, 64, 2):
        queries[i] = db['emb'][own[i]] + 0.05 * rng.standard_normal(64).astype(np.float32)
    queries /= np.linalg.norm(queries, axis=1, keepdims=True)
    queries = queries.astype(np.float32)
    # patch names of the query chunk, padded extents with context 8 (dataset/scene.py:152-160,169-171)
    ext = SceneHandler.get_extents_for_size([64, 64, 64], 16, 8, 16)
    patch_names = [SceneHandler.get_name_from_extent(scene_names[q_scene], ext[i]) for i in range(ext.shape[0])]
    database = np.concatenate([db['meta'].astype(np.float32), db['emb']], axis=1)
    with tempfile.TemporaryDirectory() as td:
        tree = Path(td)
        np.save(tree / 'database', database)
        (tree / 'index.json').write_text(json.dumps(scene_names))
        (tree / 'params.json').write_text(json.dumps({'checks': 32}))
        res_train, res_val = dict.fromkeys(patch_names), dict.fromkeys(patch_names)
        with mock.patch('builtins.print'):
            ref_ret.flann_knn_worker(res_train, K, tree, [scene_names[q_scene]] * 64, patch_names, queries, True)
            ref_ret.flann_knn_worker(res_val, K, tree, [scene_names[q_scene]] * 64, patch_names, queries, False)
        map_train = np.stack([res_train[n] for n in patch_names])       # [64,K,8]
        map_val = np.stack([res_val[n] for n in patch_names])
        # one sentinel hit so the idx<0 branch of compose is exercised
        map_val_s = map_val.copy()
        map_val_s[5, 1, :7] = database[-1, :7]
        ds_train = _FakeDataset(db['volumes'], scene_names, trunc_t, {})
        lookup = {scene_names[q_scene]: patch_names}
        ds = _FakeDataset(db['volumes'], scene_names, trunc_t, lookup)
        composed = {}
        for tag, m in (('train', map_train), ('val', map_val_s)):
            mp = {n: m[i] for i, n in enumerate(patch_names)}
            composed[tag] = ref_ret.create_retrieval_from_mapping(scene_names[q_scene], mp, K, ds_train, ds, tree).numpy()
        # query-side occupancy filter (dataset/patched_scene_dataset.py:28-32): patches below the occupancy threshold are never
        # queried -- they are absent from patch_from_scene_lookup AND from the mapping -- and keep the trunc init (:148,151)
        keep = rng.random(64) > 0.35
        keep[[0, 63]] = [False, True]
        kept_names = [n for i, n in enumerate(patch_names) if keep[i]]
        ds_masked = _FakeDataset(db['volumes'], scene_names, trunc_t, {scene_names[q_scene]: kept_names})
        mp = {n: map_val_s[i] for i, n in enumerate(patch_names) if keep[i]}
        composed['masked'] = ref_ret.create_retrieval_from_mapping(scene_names[q_scene], mp, K, ds_train, ds_masked, tree).numpy()
    save_fixture('retrieval_map_compose', seed=seed, n_patches=n_patches, q_scene=q_scene,
                        patch_keep=keep, compose_masked_sha=sha(composed['masked']), compose_masked_sub=composed['masked'][:, ::4, ::4, ::4],
                        queries=queries, db_sha=sha(db['meta'], db['emb'], db['volumes']),
                        map_train=map_train, map_val=map_val, map_val_sentinel=map_val_s,
                        compose_train_sha=sha(composed['train']), compose_val_sha=sha(composed['val']),
                        compose_train_sub=composed['train'][:, ::4, ::4, ::4], compose_val_sub=composed['val'][:, ::4, ::4, ::4],
                        extents_64_16_8_16=ext, extents_8_2_1_2=SceneHandler.get_extents_for_size([8, 8, 8], 2, 1, 2))
    print('retrieval fixture', map_train.shape, composed['train'].shape)


def gen_compose_overlap_fixture(seed=11):
    """create_retrieval_from_mapping on an OVERLAPPING patch grid (stride 8 < patch 16: 7^3 patches per 64^3 scene; dataset.no_overlap False, the branch at
    util/retrieval.py:156 decides by the mean of the stored distances): random database boxes and distances, a few sentinel hits, some patches visited with equal
    distances.  The margin between every box mean and the distance it is compared with is asserted (the reference means in float32, the restatement and the
    device in float64)."""
    ref_ret = import_reference_util_retrieval()
    from dataset.scene import SceneHandler
    cfg = rf_configs.get_config('C1')
    _, trunc_t = rf_configs.truncations(cfg)
    K = cfg['K']
    db = synthetic.make_database(seed, cfg, 64 * 6)
    scene_names = ['scene%03d' % i for i in range(db['n_scenes'])]
    rng = np.random.default_rng(seed)
    ext = SceneHandler.get_extents_for_size([64, 64, 64], 16, 8, 8)            # padded extents, context 8
    P = ext.shape[0]
    q_scene = 2
    names = [SceneHandler.get_name_from_extent(scene_names[q_scene], ext[i]) for i in range(P)]
    mapping = np.zeros((P, K, 8), dtype=np.float32)
    pick = rng.integers(0, db['meta'].shape[0] - 1, size=(P, K))
    mapping[:, :, :7] = db['meta'][pick].astype(np.float32)
    # distances up to 60: a box's last octant is never covered by an earlier patch of this grid (its stored distances are the initial 100), so means stay >= 12.5 --
    # both outcomes of the comparison at :156 occur only with distances on that scale
    mapping[:, :, 7] = rng.uniform(0.05, 60.0, size=(P, K)).astype(np.float32)
    mapping[rng.integers(0, P, 12), rng.integers(0, K, 12), 0] = -1.0           # sentinel hits (:157-158)
    mapping[40:60, :, 7] = np.float32(20.0)                                     # a run of equal distances
    ds_train = _FakeDataset(db['volumes'], scene_names, trunc_t, {})
    ds = _FakeDataset(db['volumes'], scene_names, trunc_t, {scene_names[q_scene]: names})
    ds.no_overlap = False
    with tempfile.TemporaryDirectory() as td:
        tree = Path(td)
        (tree / 'index.json').write_text(json.dumps(scene_names))
        got = ref_ret.create_retrieval_from_mapping(scene_names[q_scene], {n: mapping[i] for i, n in enumerate(names)}, K, ds_train, ds, tree).numpy()
    boxes = np.stack([np.asarray(ds_train.unpad(*ext[i].tolist()), dtype=np.int32) for i in range(P)])
    # margin of every decision, replayed in float64
    dist = np.full((K, 64, 64, 64), 100.0, dtype=np.float32)
    margin, taken = np.inf, 0
    for k in range(K):
        for p in range(P):
            b = boxes[p]
            m = dist[k, b[0]:b[1], b[2]:b[3], b[4]:b[5]].astype(np.float64).mean()
            if m != float(mapping[p, k, 7]):
                margin = min(margin, abs(m - float(mapping[p, k, 7])) / max(m, 1e-9))
            if m > float(mapping[p, k, 7]):
                dist[k, b[0]:b[1], b[2]:b[3], b[4]:b[5]] = mapping[p, k, 7]
                taken += 1
    assert margin > 1e-5, margin
    from oracle import refpath
    want = refpath.compose_retrieval_overlap(mapping, boxes, db['volumes'], K, trunc_t)
    assert np.array_equal(want, got), 'oracle restatement differs from the reference'
    save_fixture('compose_overlap', seed=seed, q_scene=q_scene, n_db_patches=64 * 6, mapping=mapping, boxes=boxes, extents=ext, taken=taken, visited=P * K,
                 min_relative_margin=margin, db_sha=sha(db['meta'], db['emb'], db['volumes']), composed_sha=sha(got), composed_sub=got[:, ::4, ::4, ::4])
    print('compose-overlap fixture', got.shape, 'copies', taken, 'of', P * K, 'margin %.2e' % margin)


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(8)
    ref_model = import_reference_model()
    gen_network_fixture(ref_model, 'net_C1', 'C1', seed=1, batch=1, with_features=True)
    gen_network_fixture(ref_model, 'net_C2_stress_b2', 'C2', seed=2, batch=2, stress=True)
    gen_network_fixture(ref_model, 'net_C3', 'C3', seed=3, batch=1)
    gen_network_fixture(ref_model, 'net_C4', 'C4', seed=4, batch=1)
    gen_network_fixture(ref_model, 'net_C5', 'C5', seed=5, batch=1, with_features=True)
    for c in ('C1', 'C4', 'C5'):
        gen_query_fixture(ref_model, c, seed=6)
    for c in ('C1', 'C5'):
        gen_dbrow_fixture(ref_model, c)
    gen_retrieval_fixture()
    gen_compose_overlap_fixture()
    gen_combine_fixture()
    gen_loss_fixture()



# ------------------------------------------------------------------------- "next" row N3: scene recomposition
def gen_combine_fixture(seed=9):
    """PatchedSceneDataset.combine_chunks / combine_inputs / combine_targets (dataset/patched_scene_dataset.py:153-186) run
    unchanged on an object that carries just the attributes they touch."""
    import_reference_util_retrieval()                       # installs the stubs dataset.* needs
    from dataset.patched_scene_dataset import PatchedSceneDataset
    out = {'seed': seed}
    for tag, cfg_name, names in (
            ('front', 'C3', ['sceneA__room0__0_0_0', 'sceneA__room0__64_0_0', 'sceneA__room0__64_128_64', 'sceneB__room3__0_64_0', 'sceneB__room3__0_0_0']),
            ('shapenet', 'C1', ['03001627_aaa', '04379243_bbb'])):
        cfg = rf_configs.get_config(cfg_name)
        trunc_i, trunc_t = rf_configs.truncations(cfg)
        chunks = [synthetic.make_chunk(seed * 100 + i, cfg) for i in range(len(names))]
        ds = object.__new__(PatchedSceneDataset)
        ds.scenes = names
        ds.dataset_name = cfg['dataset_train']['dataset_name']
        tgt = {n: c['target_raw'] for n, c in zip(names, chunks)}
        inp = {n: c['input_raw'] for n, c in zip(names, chunks)}
        s_in = cfg['dataset_train']['input_chunk_size']
        res_t = ds.combine_chunks(1, 64, trunc_t, lambda obj, n: obj[n], tgt)                       # combine_targets, :179-180
        res_i = ds.combine_chunks(64 / s_in, s_in, trunc_i, lambda obj, n: obj[n], inp)             # combine_inputs, :176-177
        # the inference loop's recomposition (trainer/train_refinement.py:166): combine_retrievals(predictions [n, 1, 64^3] as float16, 0)
        preds = np.stack([synthetic.uniform_stress_volume(seed * 1000 + i, (64, 64, 64), trunc_t) for i in range(len(names))])[:, None].astype(np.float16)
        ds.scene_handler = types.SimpleNamespace(target_chunk_size=64, target_trunc=trunc_t)      # what the two properties read (:44-45, :72-73)
        res_p = ds.combine_retrievals(preds, 0)
        for k in sorted(res_p):
            out['%s_pred_%s_shape' % (tag, k)] = np.array(res_p[k].shape)
            out['%s_pred_%s_sha' % (tag, k)] = sha(res_p[k])
        out[tag + '_names'] = np.array(names)
        out[tag + '_cfg'] = cfg_name
        out[tag + '_keys'] = np.array(sorted(res_t))
        for k in sorted(res_t):
            out['%s_target_%s_shape' % (tag, k)] = np.array(res_t[k].shape)
            out['%s_target_%s_sha' % (tag, k)] = sha(res_t[k])
            out['%s_target_%s_dtype' % (tag, k)] = str(res_t[k].dtype)
            out['%s_input_%s_shape' % (tag, k)] = np.array(res_i[k].shape)
            out['%s_input_%s_sha' % (tag, k)] = sha(res_i[k])
    save_fixture('combine_chunks', **out)
    print('combine fixture', [k for k in out if k.endswith('_shape')])


# ---------------------------------------------------------------------------------- "next" row N1: database build
def gen_dbrow_fixture(ref_model, cfg_name='C1', seed=8):
    """DB-side rows as create_dictionary builds them (util/retrieval.py:29-45): 32^3 target windows (16 + 8 context, padded
    with trunc, dataset/scene.py:71,94) normalised (dataset/patched_scene_dataset.py:128) -> fenc_target -> L2 normalise;
    plus the 'zero patch' row: an all-ONES 32^3 patch fed un-normalised (util/retrieval.py:21-26)."""
    cfg = rf_configs.get_config(cfg_name)
    _, trunc_t = rf_configs.truncations(cfg)
    _, fenc_target = ref_model.get_retrieval_networks(cfg['retrieval_model'])
    load_seeded(fenc_target, seed * 1000 + 22)
    raw = synthetic.make_chunk(seed * 100, cfg)['target_raw']
    g = cfg['query_geometry']
    ps, pc = g['patch_size_target'], g['patch_context_target']
    padded = np.pad(raw, pc, mode='constant', constant_values=trunc_t)
    w = ps + 2 * pc
    wins = np.stack([padded[x:x + w, y:y + w, z:z + w] for x in range(0, 64, ps) for y in range(0, 64, ps) for z in range(0, 64, ps)])[:, None]
    wins = synthetic.normalise_target(cfg, wins)
    lat = cfg['retrieval_model']['latent_dim']
    with torch.no_grad():
        emb = torch.nn.functional.normalize(fenc_target(torch.from_numpy(wins)).permute((0, 2, 3, 4, 1)).reshape((-1, lat)), dim=1).numpy()
        ones = torch.from_numpy(np.ones([w] * 3, dtype=np.float32)).unsqueeze(0).unsqueeze(0)
        zero_emb = torch.nn.functional.normalize(fenc_target(ones).permute((0, 2, 3, 4, 1)).reshape((-1, lat)), dim=1).numpy()
    zero_row = np.hstack([np.array([-1], dtype=np.float32)[:, np.newaxis]] + [np.array([0], dtype=np.float32)[:, np.newaxis], np.array([ps], dtype=np.float32)[:, np.newaxis]] * 3 + [zero_emb])
    save_fixture('dbrow_%s' % cfg_name, cfg_name=cfg_name, seed=seed, windows_sha=sha(wins), emb=emb, zero_row=zero_row)
    print('dbrow', cfg_name, wins.shape, emb.shape, zero_row.shape)


# ------------------------------------------------------------------------- model.loss (imported by the reference's trainers)
def gen_loss_fixture(seed=11):
    """NT-Xent (cosine / dot, with and without the IoU-dependent temperature), Gram-matrix style loss and the normal cosine similarity
    from the reference's own model/loss.py.  Its forward pins the mask with ``.cuda(device)`` (model/loss.py:57,62), which refuses a CPU
    device: for this call only, Tensor.cuda is mapped to Tensor.to."""
    import importlib
    ref_loss = importlib.import_module('model.loss')
    assert str(REF) in ref_loss.__file__, ref_loss.__file__
    g = torch.Generator().manual_seed(seed)
    b, c = 6, 16
    zis = torch.randn(b, c, generator=g)
    zjs = zis + 0.3 * torch.randn(b, c, generator=g)
    iou = torch.rand(2 * b, 2 * b, generator=g)
    pn = torch.randn(2, 3, 4, 4, 4, generator=g)
    tn = torch.randn(2, 3, 4, 4, 4, generator=g)
    pn[:, :, 0] = 0
    tn[:, :, :, 1] = 0
    out = {'zis': zis.numpy(), 'zjs': zjs.numpy(), 'iou': iou.numpy(), 'pred_norms': pn.numpy(), 'target_norms': tn.numpy()}
    with mock.patch.object(torch.Tensor, 'cuda', lambda self, device=None, **kw: self.to(device)):
        for cos in (True, False):
            for temp in (0.5, 0.07):
                crit = ref_loss.NTXentLoss(temp, cos)
                key = '%s_t%g' % ('cos' if cos else 'dot', temp)
                zi = zis.clone().requires_grad_(True)
                zj = zjs.clone().requires_grad_(True)
                loss = crit(zi, zj)
                loss.backward()
                out['loss_' + key] = np.float64(loss.item())
                out['grad_zis_' + key] = zi.grad.numpy()
                out['grad_zjs_' + key] = zj.grad.numpy()
                out['loss_iou_' + key] = np.float64(crit(zis, zjs, iou).item())
    out['style'] = np.float64(ref_loss.patch_style_loss(zis, zjs).item())
    out['normal_cos'] = np.float64(ref_loss.get_cosine_similarity(pn, tn).item())
    save_fixture('loss', **out)
    print('loss:', {k: float(v) for k, v in out.items() if np.ndim(v) == 0})


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'loss':
        import_reference_model()
        gen_loss_fixture()
    elif len(sys.argv) > 1 and sys.argv[1] == 'combine':
        import_reference_model()
        gen_combine_fixture()
    else:
        main()
